"""bench.py's N > 1 flow (rank bookkeeping, pipelined all-gather, fences, max-over-ranks timing, rank-0 JSON) on
ONE GPU: two ranks sharing cuda:0 over the gloo backend.  Throughput is meaningless here; what is checked is that
the multi-rank code path runs end to end on real device tensors and reports a well-formed line."""
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


@pytest.mark.parametrize("extra", [[], ["--sync-gather"], ["--no-gather"]])
def test_two_rank_bench_line(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "4", "--backend", "gloo", "--single-device", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3 * 1e-3)) / d["value"] < 1e-3
    assert ("all_gather" in d["config"]["parallelism"]) == ("--no-gather" not in extra)
    assert "roofline" in d and "cpu_baseline" not in d
    assert ("multi_gpu" in d) == ("--no-gather" not in extra)
    if "multi_gpu" in d:
        mg = d["multi_gpu"]
        assert mg["forward_only_ms"] > 0 and mg["gather_only_ms"] > 0
        assert mg["gather_bytes_received_per_rank"] == 4 * 144 * 4096 * 2
