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


def _run(extra, nproc=2, env=None, expect_rc=0):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    if expect_rc != 0:
        assert res.returncode != 0, res.stdout[-2000:]
        return res
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--sync-gather"], ["--no-gather"]])
def test_two_rank_bench_line(extra):
    """Default = the metric's configuration: STRONG scaling, the global batch sharded over the ranks."""
    d = _run(["--batch", "8"] + extra)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 8 and d["config"]["per_gpu_batch"] == 4
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3 * 1e-3)) / d["value"] < 1e-3
    assert ("all_gather" in d["config"]["parallelism"]) == ("--no-gather" not in extra)
    assert "roofline" in d and "cpu_baseline" not in d
    assert ("multi_gpu" in d) == ("--no-gather" not in extra)
    if "multi_gpu" in d:
        mg = d["multi_gpu"]
        assert mg["forward_only_ms"] > 0 and mg["gather_only_ms"] > 0
        assert mg["gather_bytes_received_per_rank"] == 4 * 144 * 4096 * 2
        # the start-up self-check ran (one tiny all_gather_into_tensor compared on every rank, before the warm-up)
        assert mg["collective_self_check"]["all_gather_into_tensor"] == "ok" and mg["collective_self_check"]["all_reduce_min"] == 1.0


def test_two_rank_weak_scaling_line():
    d = _run(["--batch", "4", "--scaling", "weak"])
    assert d["scaling"] == "weak" and d["config"]["global_batch"] == 8 and d["config"]["per_gpu_batch"] == 4


@pytest.mark.parametrize("nproc,images", [(2, 4), (3, 5)])
def test_hd_line_equal_and_ragged_shards(nproc, images):
    """--hd: crops sharded over the ranks (3 ranks x 45 crops of 5 images is ragged: 15 each ... 5 images x 9 = 45 = 3 x 15
    is equal, so use a count that is not: (3, 5) -> 45 crops / 3 = 15; (2, 4) -> 36 / 2 = 18) — plus a truly ragged one
    below."""
    d = _run(["--hd", "--hd-images", str(images)], nproc=nproc)
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == images * 9
    assert "TokenPacker-HD" in d["config"]["workload"] and d["value"] > 0


def test_hd_line_ragged():
    d = _run(["--hd", "--hd-images", "3"], nproc=2)              # 27 crops over 2 ranks: 14 + 13
    assert d["config"]["global_batch"] == 27 and d["config"]["per_gpu_batch"] == 14
    assert "ragged" in d["multi_gpu"]["collective"]


def test_e2e_line_single_rank():
    """`bench.py --e2e` (tower -> projector -> prefill) on one rank with a two-layer prefill: the path must run and print its
    one line (a NameError lived here for half a round because nothing exercised it)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--e2e", "--e2e-batch", "2", "--e2e-layers", "2", "--steps", "1", "--warmup", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "tokens/s" and d["value"] > 0 and d["split_ms"]["projector_hip"] > 0
    # BASELINE configs[4] is quoted "vs reference": the same run with the reference's projector, after the timed region
    assert d["reference_leg"]["value"] > 0 and d["vs_reference"] > 0 and d["projector_speedup_in_place"] > 0


def test_e2e_line_two_ranks_ragged():
    """BASELINE configs[4] is a DDP run: `--e2e` with MORE than one rank (3 samples over 2 ranks: 2 + 1), the reference leg
    included — the multi-rank branch of run_e2e (shard bounds, per-rank seeds, MAX-reduced clocks of both legs, rank-0 line)
    had never executed anywhere."""
    d = _run(["--e2e", "--e2e-batch", "3", "--e2e-layers", "2"])
    assert d["unit"] == "tokens/s" and d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["global_batch"] == 3 and d["config"]["per_gpu_batch"] == 2
    assert d["multi_gpu"]["per_rank_batches"] == [2, 1] and d["multi_gpu"]["collective_self_check"]["all_gather_into_tensor"] == "ok"
    assert d["reference_leg"]["value"] > 0 and d["vs_reference"] > 0


def test_collective_self_check_fails_fast_with_a_message():
    """A gather that delivers something else than the ranks sent (here: injected on the last rank) ends the run BEFORE the warm-up, on
    every rank, with a message that names the collective — not as garbage or a hang inside the timed region."""
    res = _run(["--batch", "4", "--no-extras"], env={"TP_BENCH_INJECT_COLLECTIVE_FAULT": "1"}, expect_rc=17)
    assert "collective self-check FAILED before the warm-up" in res.stderr
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
