#!/usr/bin/env python3
"""Where a K-tile of the ping-pong kernel's main loop goes: timing probes (TP_TUNE_PAIR_DEBUG >> 4: builds of gemm8_kernel that
leave out part of the loop — results are garbage, timings are not) on a fp16 36864 x 4096 x K launch, K = 1024 and 4096, arms
interleaved; per probe the least-squares line  t = fixed + per_ktile * K / 64  per tile (9 tile rounds per CU).

    python tools/loop_probe.py [--out gpurun_out/loop_probe.json]
"""
import os as _os
_os.environ.setdefault("TP_LIB_VARIANT", "exp")     # the timing-probe instantiations live in libtokenpacker_exp.so only (make exp)
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenpacker_amd import _capi  # noqa: E402
from tests import gpu_util as gu  # noqa: E402

PROBES = [(0, "all on"), (1, "no b0 reads (phase 0: 8 instead of 12)"), (3, "no W reads (phases 0 / 1)"), (15, "no fragment reads"),
          (16, "no DMA in the loop"), (31, "no reads, no DMA"), (64, "no MFMAs"),
          (79, "no MFMAs, no fragment reads (DMA + barriers)"), (80, "no MFMAs, no DMA (fragment reads + barriers)")]


def rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/loop_probe.json")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reserve", type=int, default=0, help="TP_TUNE_RESERVE_CUS: run on (32 - reserve) CUs per XCD (16 / 24 / 28: 128 / 64 / 32 CUs; M is scaled to keep 9 tile rounds per CU)")
    ap.add_argument("--rows", type=int, default=0, help="M (default 36864 = 9 tile rounds per CU; 20480 = 5 rounds, A = 168 MB at K = 4096: inside the Infinity Cache)")
    a = ap.parse_args()
    assert (32 - a.reserve) % 4 == 0 or a.reserve == 0, "reserve: 0, 4, 8, ..., 28 (whole rounds)"
    M, N = (a.rows or 36864) * (32 - a.reserve) // 32, 4096
    rounds_per_cu = M // 256 * 16 / ((32 - a.reserve) * 8)
    _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, a.reserve)
    ops = {}
    for K in (1024, 4096):
        ops[K] = (rand((M, K), torch.float16, 1), rand((N, K), torch.float16, 2, K ** -0.5), rand((N,), torch.float32, 3))
    _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 1)
    times = {(p, K): [] for p, _ in PROBES for K in ops}
    try:
        for p, _ in PROBES:                       # warm-up
            _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, p << 4)
            for K, (A, W, b) in ops.items():
                gu.linear(A, W, bias=b, out_dtype=torch.float16, tile=256)
        for _ in range(a.rounds):
            for p, _ in PROBES:
                _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, p << 4)
                for K, (A, W, b) in ops.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(5):
                        gu.linear(A, W, bias=b, out_dtype=torch.float16, tile=256, sync=False)
                    e1.record()
                    torch.cuda.synchronize()
                    times[(p, K)].append(e0.elapsed_time(e1) / 5)
    finally:
        _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 0)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)
        _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, 0)
    res = []
    for p, name in PROBES:
        t1, t4 = statistics.median(times[(p, 1024)]), statistics.median(times[(p, 4096)])
        per = (t4 - t1) / rounds_per_cu / 48 * 1e3
        r = {"probe": p, "what": name, "ms_K1024": round(t1, 4), "ms_K4096": round(t4, 4), "us_per_ktile": round(per, 4),
             "us_fixed_per_tile": round(t1 / rounds_per_cu * 1e3 - 16 * per, 3), "us_per_tile_K4096": round(t4 / rounds_per_cu * 1e3, 2)}
        res.append(r)
        print(r, flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"shape": [M, N], "cus": (32 - a.reserve) * 8, "probes": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
