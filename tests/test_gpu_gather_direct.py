"""shard.DirectGather (tp_gather_*: the all-gather of projected tokens as per-peer copies + sequence flags, no collective
kernel) between PROCESSES, on the one GPU the box has: two and three ranks all driving cuda:0 through HIP IPC, uneven load,
equal and ragged shards, depth 2 and 3, every step compared bit for bit with the expectation; and bench.py's N > 1 flow
started WITHOUT a launcher (`python bench.py --gpus 2`), in both gather modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**kw):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(kw)
    return env


@pytest.mark.parametrize("nproc,use_cus", [(2, 0), (3, 0), (2, 1)])
def test_direct_gather_between_processes(nproc, use_cus):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "gather_worker.py")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env(TP_GATHER_USE_CUS=str(use_cus)))
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert res.stdout.count("gather scenarios OK") == nproc


def _bench_no_launcher(extra):
    """`python bench.py --gpus 2 ...` exactly as the driver types it for N > 1, with no torch.distributed.run around it."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--single-device", "--no-cpu-baseline", "--batch", "8", "--min-seconds", "0.05"] + extra
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["rccl", "sdma"])
def test_bench_spawns_its_own_ranks(mode):
    d = _bench_no_launcher(["--gather", mode, "--probe-other-gather"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["per_gpu_batch"] == 4
    mg = d["multi_gpu"]
    assert mg["gather"] == mode and mg["ranks"] == 2 and mg["backend"] == "gloo"
    assert [r["rank"] for r in mg["rank_devices"]] == [0, 1] and len({r["pid"] for r in mg["rank_devices"]}) == 2
    assert mg["forward_only_ms"] > 0 and mg["gather_only_ms"] > 0
    assert mg["other_gather"]["mode"] == ("sdma" if mode == "rccl" else "rccl") and mg["other_gather"]["gather_only_ms"] > 0
    assert d["timing"]["long_run"]["blocks"] >= 5


def test_bench_auto_proves_the_sdma_transport_then_times_both():
    """The default (--gather auto): the copy-engine transport is self-tested (every rank checks every peer's rows), both
    transports are timed over the K steps with the same protocol, the faster one is the line's value, both are reported."""
    d = _bench_no_launcher([])
    mg = d["multi_gpu"]
    assert mg["gather_requested"] == "auto" and mg["sdma_self_test"] == "passed"
    assert mg["sdma_flags_fine_grained"] is True         # the flags the sync kernel polls are fine-grained device memory
    assert set(mg["transports_timed_ms_per_step"]) == {"rccl", "sdma"}
    best = min(mg["transports_timed_ms_per_step"], key=mg["transports_timed_ms_per_step"].get)
    assert mg["gather"] == best and abs(d["ms_per_step"] - mg["transports_timed_ms_per_step"][best]) < 1e-3
    assert mg["other_gather"]["mode"] == ("sdma" if best == "rccl" else "rccl") and mg["other_gather"]["gather_only_ms"] > 0


def test_bench_sdma_sync_gather_and_ragged():
    d = _bench_no_launcher(["--gather", "sdma", "--sync-gather", "--batch", "7"])
    assert d["config"]["global_batch"] == 7 and d["config"]["per_gpu_batch"] == 4 and d["multi_gpu"]["gather"] == "sdma"
