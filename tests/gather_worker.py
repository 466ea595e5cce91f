"""Worker of tests/test_gpu_gather_direct.py: one rank of a world_size-N group whose ranks ALL drive cuda:0 (the GPU box has
one GPU; HIP IPC between processes on one device exercises the same export / open / push / flag path as between devices).
Every rank can compute every rank's shard of every step from (rank, step), so the expected gathered tensor needs no
collective.  Exit code 0 = every step of every scenario was bit-identical to the expectation."""
import os
import random
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard_values(rank, step, rows, tail, dtype, device):
    g = torch.Generator(device="cpu").manual_seed(1000003 * step + 101 * rank + 7)
    return torch.randn((rows,) + tuple(tail), generator=g, dtype=torch.float32).to(dtype).to(device)


def main():
    from tokenpacker_amd import _capi, shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _capi.load_library()
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    rng = random.Random(17 + rank)
    use_cus = int(os.environ.get("TP_GATHER_USE_CUS", "0"))
    for total, tail, depth, steps in ((8, (144, 256), 3, 24), (7, (36, 128), 3, 16), (8, (64, 512), 2, 16), (1, (16, 128), 3, 6)):
        sizes = shard.shard_sizes(total, world)
        g = shard.DirectGather(total, tail, torch.bfloat16, dev, depth=depth, use_cus=bool(use_cus), timeout_ms=20000)
        lag = depth - 2                                  # how many submits a result may trail by (the class's contract)
        tickets = []

        def verify(step):
            buf = g.result(tickets[step])
            exp = torch.cat([shard_values(r, step, sizes[r], tail, torch.bfloat16, dev) for r in range(world)], dim=0)
            assert torch.equal(buf, exp), f"rank {rank}: total {total} depth {depth} step {step} differs"

        for i in range(steps):
            # uneven load: one rank or the other is late by a host sleep and / or by a kernel that holds CUs on its stream
            if rng.random() < 0.4:
                time.sleep(rng.random() * 0.01)
            if rng.random() < 0.4:
                _capi.check(_capi.load_test_library().tp_test_occupy_cus(8, 200 + int(rng.random() * 2000), sink.data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream), "occupy")
            view = g.begin()
            mine = shard_values(rank, i, sizes[rank], tail, torch.bfloat16, dev)
            if i % 2 == 0:
                view.copy_(mine)                         # the projector's _out path: the shard is produced in place
                tickets.append(g.submit())
            else:
                tickets.append(g.submit(mine))           # a shard that lives elsewhere is copied in
            if i - lag >= 0:
                verify(i - lag)
        for i in range(max(steps - lag, 0), steps):
            verify(i)
        g.close()
    # a peer that never submits: the wait gives up after its timeout and close() reports it (no hang)
    g = shard.DirectGather(4, (8, 128), torch.bfloat16, dev, depth=3, timeout_ms=300)
    rows = shard.shard_sizes(4, world)[rank]
    if rank == 0:
        t = g.submit(torch.zeros(rows, 8, 128, dtype=torch.bfloat16, device=dev))
        g.result(t)
        torch.cuda.synchronize(dev)
        try:
            g.check()
            raise SystemExit("rank 0: the timeout was not reported")
        except TimeoutError:
            pass
    dist.barrier()
    if rank != 0:                                        # now catch up so that close() finds a consistent state
        g.submit(torch.zeros(rows, 8, 128, dtype=torch.bfloat16, device=dev))
    g.status.zero_()
    g.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: gather scenarios OK", flush=True)


if __name__ == "__main__":
    main()
