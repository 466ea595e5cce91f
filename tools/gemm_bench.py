#!/usr/bin/env python3
"""Micro-benchmark of the tp_linear MFMA kernels on the GEMM shapes of the TokenPacker path (run on the
GPU box).  Variants are interleaved round-robin inside ONE process (guide §5.4 rule 24) on random normal
data (rule 25); prints median TFLOP/s per (shape, variant) and, as a yard-stick, torch.matmul
(hipBLASLt/rocBLAS) on the same operands.

    python tools/gemm_bench.py [--batch 256] [--scale-factor 2] [--rounds 7]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import _capi  # noqa: E402

G, S, F = _capi.TP_LINEAR_GELU, _capi.TP_LINEAR_ROW_STATS, _capi.TP_LINEAR_LN_FOLD
# (name, forced tile, TP_TUNE_PAIR_GEMM, TP_TUNE_PAIR_STAGGER)
VARIANTS = [("tile128", 128, 1, 100), ("pp_persistent", 256, 1, 100), ("pair", 0, 2, 100), ("pair_nostagger", 0, 2, 0)]


def shapes(B, s, D):
    N, M = 576, (24 // s) ** 2
    return [  # name, M, N, K, flags, in dtype, out dtype
        ("kv_layer0", B * N, 2048, 4096, G, torch.bfloat16, _capi.TP_F16),
        ("kv_layer2", B * N, 1024, 1024, S, torch.float16, _capi.TP_F16),
        ("kv_inproj", B * N, 1024, 1024, F, torch.float16, _capi.TP_F16),
        ("q_side", B * M, 1024, 1024, 0, torch.float16, _capi.TP_F16),
        ("mlp0", B * M, D, 1024, G, torch.float16, _capi.TP_F16),
        ("mlp2", B * M, D, D, 0, torch.float16, _capi.TP_BF16),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--scale-factor", type=int, default=2)
    ap.add_argument("--hidden-size", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/gemm_bench.json")
    args = ap.parse_args()
    lib = _capi.load_library()
    st = torch.cuda.current_stream().cuda_stream
    results = []
    for name, M, N, K, flags, dt, odt in shapes(args.batch, args.scale_factor, args.hidden_size):
        A = torch.randn(M, K, device="cuda", dtype=torch.float32).to(dt)
        W = (torch.randn(N, K, device="cuda", dtype=torch.float32) * K ** -0.5).to(dt)
        bias = torch.randn(N, device="cuda")
        colsum = torch.randn(N, device="cuda")
        mr = torch.rand(M, 2, device="cuda") + 0.5
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
        stats = torch.empty(8 * M * 2, device="cuda")
        fl = 2.0 * M * N * K

        def make(tile, pair, stagger):
            a = _capi.tp_linear_args()
            a.M, a.N, a.K, a.flags = M, N, K, flags
            a.dtype = _capi.TP_BF16 if dt == torch.bfloat16 else _capi.TP_F16
            a.out_dtype = odt
            a.lda, a.ldc, a.tile = K, N, tile
            a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr()
            a.row_stats_out, a.row_mean_rstd, a.colsum = stats.data_ptr(), mr.data_ptr(), colsum.data_ptr()

            def run():
                _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, pair)
                _capi.set_tuning(_capi.TP_TUNE_PAIR_STAGGER, stagger)
                rc = lib.tp_linear(ctypes.byref(a), st)
                assert rc == 0, _capi.last_error()
            return run

        runs = [(v, make(t, pr, sg)) for v, t, pr, sg in VARIANTS if t == 0 or N % t == 0]
        runs.append(("torch.matmul", lambda: torch.matmul(A, W.t())))
        times = {v: [] for v, _ in runs}
        for v, fn in runs:                       # warm-up
            fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v, fn in runs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.iters)
        for v, _ in runs:
            ms = statistics.median(times[v])
            rec = dict(shape=name, M=M, N=N, K=K, variant=v, ms=round(ms, 4), ms_min=round(min(times[v]), 4),
                       tflops=round(fl / ms / 1e9, 1))
            results.append(rec)
            print(rec, flush=True)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_STAGGER, 100)
        del A, W, C, mr, stats
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
