"""Multi-GPU path on CPU: world_size-2 gloo processes, batch sharded, ONE
all-gather.  The HIP projector cannot run here, so a stand-in with the projector's forward
contract (the oracle, fp32) is injected into tokenpacker_amd.shard.project_sharded."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tokenpacker_amd import shard, synth


def test_shard_bounds_cover_and_balance():
    for total in (1, 7, 8, 36, 288, 289, 295):
        for ws in (1, 2, 3, 4, 8):
            spans = [shard.shard_bounds(total, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == shard.shard_sizes(total, ws)
    assert shard.shard_sizes(288, 8) == [36] * 8          # BASELINE config 4: 32 images x 9 crops
    with pytest.raises(ValueError):
        shard.shard_bounds(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, chunks, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import tokenpacker_oracle as orc
        torch.set_num_threads(2)
        params = synth.make_params(2, 256)
        x, xm = synth.make_inputs(3, total)

        def project(pair):
            return orc.forward(params, pair[0], pair[1], scale_factor=4, compute_dtype=torch.float32)

        xl, xml = shard.local_shard(x), shard.local_shard(xm)
        y = shard.project_sharded(project, xl, xml, total, overlap_chunks=chunks)
        y_ref = project((x, xm))
        ok = y.shape == y_ref.shape and torch.allclose(y, y_ref, atol=1e-5)
        # every rank must hold the full, identically ordered result
        gathered = [torch.empty_like(y) for _ in range(world)]
        dist.all_gather(gathered, y)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        # no-gather mode returns only the local shard
        y_loc = shard.project_sharded(project, xl, xml, total, gather=False)
        lo, hi = shard.shard_bounds(total, world, rank)
        ok_loc = torch.allclose(y_loc, y_ref[lo:hi], atol=1e-5)
        # pipelined gather of a stream of batches: two batches in flight, results in rank order
        pipe = shard.TokenGatherPipeline(total, depth=2) if total % world == 0 else None
        ok_pipe = True
        if pipe is not None:
            outs = []
            for k in range(3):
                slot = pipe.submit(y_loc * (k + 1))
                outs.append((slot, k))
                if k >= 1:                       # consume the batch submitted one step earlier
                    ps, pk = outs[k - 1]
                    ok_pipe &= torch.equal(pipe.result(ps), y * (pk + 1))
            pipe.drain()
            ok_pipe &= torch.equal(pipe.result(outs[-1][0]), y * 3)
        # ragged shards, addressed IN PLACE: one all_gather_into_tensor of b_max-row slots, no pad / cat copies.  The
        # stand-in advertises the HIP module's `_out` contract, so its result is written straight into the gather slot.
        ok_ragged = True
        if total % world != 0:
            class Proj:
                supports_out = True
                calls = []

                def out_like(self, xx):
                    return torch.empty(0, dtype=torch.float32), (36, 256)

                def __call__(self, pair, _out=None):
                    yy = project(pair)
                    self.calls.append(_out is not None)
                    if _out is None:
                        return yy
                    _out.copy_(yy)
                    return _out
            pj = Proj()
            g = shard.project_sharded(pj, xl, xml, total, dense=False)
            sizes = shard.shard_sizes(total, world)
            ok_ragged &= isinstance(g, shard.GatheredTokens) and pj.calls == [True] and len(g) == total
            ok_ragged &= tuple(g.buf.shape) == (world * max(sizes), 36, 256) and g.sizes == sizes
            ok_ragged &= all(torch.allclose(g[i], y_ref[i], atol=1e-5) for i in range(total))
            ok_ragged &= g.crop_map().tolist() == [g.row_of(i) for i in range(total)]
            ok_ragged &= torch.allclose(g.compact(), y_ref, atol=1e-5)
            ok_ragged &= torch.allclose(shard.project_sharded(pj, xl, xml, total), y_ref, atol=1e-5)     # dense=True: one copy
            # asynchronous ragged gather returns a REAL work handle (wait() used to crash on None)
            g2, work = shard.all_gather_tokens(y_loc, total, async_op=True, dense=False)
            work.wait()
            ok_ragged &= all(torch.allclose(g2[i], y_ref[i], atol=1e-5) for i in range(total))
            try:
                shard.all_gather_tokens(y_loc, total, async_op=True)            # dense + async on ragged shards is refused
                ok_ragged = False
            except ValueError:
                pass
        ret[rank] = bool(ok and same and ok_loc and ok_pipe and ok_ragged)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,chunks", [(4, 1), (5, 1), (4, 2), (3, 1)])
def test_two_rank_shard_and_all_gather(total, chunks):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, chunks, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
