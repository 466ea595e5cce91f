#!/usr/bin/env python3
"""Fixed per-tile cost vs main-loop cost of the persistent 256x256 GEMM: time per tile-round at K = 256 ... 4096
for each epilogue flavour, then a least-squares line  t = fixed + per_ktile * (K / 64)  (run on the GPU box)."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import _capi  # noqa: E402

G, S, F = _capi.TP_LINEAR_GELU, _capi.TP_LINEAR_ROW_STATS, _capi.TP_LINEAR_LN_FOLD


def main():
    args = dict(a.split("=", 1) for a in sys.argv[1:])           # lib=<path to a probe build> for A/B runs on one box
    lib = _capi.load_library(args["lib"]) if "lib" in args else _capi.load_library()
    half = args.get("tiles") == "half"                                # tiles=half: TP_TUNE_GEMM_TILE = 2 (128 x 256 half tiles)
    if half:
        _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, 2)
    print("args", args)
    st = torch.cuda.current_stream().cuda_stream
    M, N = 147456, 1024
    KS = (256, 512, 1024, 2048, 4096)
    bias = torch.randn(N, device="cuda")
    colsum = torch.randn(N, device="cuda")
    mr = torch.rand(M, 2, device="cuda") + 0.5
    C = torch.empty(M, N, device="cuda", dtype=torch.float16)
    stats = torch.empty(8 * M * 2, device="cuda")
    rows = {}
    for K in KS:
        A = (0.05 * torch.randn(M, K, device="cuda")).to(torch.float16)
        W = (0.05 * torch.randn(N, K, device="cuda")).to(torch.float16)
        for fname, flags in (("plain", 0), ("stats", S), ("lnfold", F), ("gelu", G)):
            a = _capi.tp_linear_args()
            a.M, a.N, a.K, a.flags = M, N, K, flags
            a.dtype, a.out_dtype = _capi.TP_F16, _capi.TP_F16
            a.lda, a.ldc, a.tile = K, N, (0 if half else 256)
            a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr()
            a.row_stats_out, a.row_mean_rstd, a.colsum = stats.data_ptr(), mr.data_ptr(), colsum.data_ptr()
            ts = []
            for r in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    assert lib.tp_linear(ctypes.byref(a), st) == 0
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            ms = statistics.median(ts[1:])
            rounds = (M // 256) * (N // 256) / 256
            us = ms * 1e3 / rounds
            rows.setdefault(fname, []).append((K, us))
            print(f"K={K:5d} {fname:7s} {ms * 1e3:8.1f} us = {us:7.2f} us per tile-round  "
                  f"{2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
        del A, W
    for fname, pts in rows.items():
        xs = [k / 64 for k, _ in pts if k >= 512]
        ys = [u for k, u in pts if k >= 512]
        n = len(xs)
        mx, my = sum(xs) / n, sum(ys) / n
        b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        print(f"fit {fname:7s}: fixed {my - b * mx:6.2f} us per tile + {b:6.3f} us per K-tile "
              f"(main loop alone = {2.0 * 256 * 256 * 64 * 256 / b / 1e6:7.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
