"""One-rank RCCL smoke: the GPU boxes of this pool have ONE device and RCCL refuses two ranks on one
device, so the multi-rank tests run over gloo.  What a one-rank `nccl` group still proves before the driver's 8-GPU run: the
RCCL library loads, `device_id=` binds the communicator, `all_gather_into_tensor` on the communicator's own stream next to the
projector's side-stream pipeline is legal, and the collective code paths (`project_sharded`, `TokenGatherPipeline`, ragged
`all_gather_tokens`) return the local forward bit for bit.  Runs in a child process (the suite's process must not own a
process group)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from tokenpacker_amd import TokenPacker, shard, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
B, s, D = 6, 2, 256
model = TokenPacker(hidden_size=D, scale_factor=s)
model.load_state_dict(synth.make_params(1, D))
model = model.to(device=dev, dtype=torch.bfloat16).eval().requires_grad_(False)
x, xm = synth.make_inputs(2, B, torch.bfloat16, layout="tower")
x, xm = x.to(dev), xm.to(dev)
with torch.no_grad():
    y_ref = model((x, xm))
    # 1. project_sharded through the collective (force_collective: a one-rank group has nothing to gather otherwise)
    y = shard.project_sharded(model, x, xm, B, force_collective=True)
    assert y.data_ptr() != y_ref.data_ptr() and torch.equal(y, y_ref), "project_sharded over a one-rank nccl group"
    # ... and the chunked form (two async gathers in flight while the second chunk is projected)
    y2 = shard.project_sharded(model, x, xm, B, overlap_chunks=2, force_collective=True)
    assert torch.equal(y2, y_ref), "overlap_chunks=2"
    # 2. the pipelined gather bench.py times: gather of step i beside the forward of step i + 1, depth 2
    pipe = shard.TokenGatherPipeline(B, depth=2)
    prev = None
    for k in range(5):
        slot = pipe.submit(model((x, xm)))
        if prev is not None:
            assert torch.equal(pipe.result(prev), y_ref), f"pipelined gather, step {k - 1}"
        prev = slot
    pipe.drain()
    assert torch.equal(pipe.result(prev), y_ref)
    # 3. the ragged form's slot path (one rank: b_max == b, still through all_gather_into_tensor)
    g = shard.all_gather_tokens(y_ref, B, dense=False)
    g = g.buf if isinstance(g, shard.GatheredTokens) else g
    assert torch.equal(g[:B], y_ref)
    # 4. a reduction as well (bench.py's clock is an all_reduce MAX)
    t = torch.tensor([3.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 3.5
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_one_rank_nccl_group_through_the_collective_code():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "RCCL_ONE_RANK_OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])


@pytest.mark.gpu
def test_bench_force_dist_line_matches_the_plain_shape():
    """`bench.py --gpus 1 --force-dist`: the N = 1 line through the collective path (nccl, one rank) — same workload, a
    `multi_gpu` block with backend nccl and the self-check that runs before the warm-up."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--batch", "8", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--min-seconds", "0"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    lines = res.stdout.strip().splitlines()
    assert lines[-1].startswith("{"), ("the JSON line must be the LAST line of stdout (RCCL's banner flushed before it)", lines[-3:])
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 8 and line["value"] > 0
    mg = line["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["ranks"] == 1 and mg["collective_self_check"]["all_gather_into_tensor"] == "ok"
    assert "all_gather" in line["config"]["parallelism"]
