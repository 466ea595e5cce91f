#!/usr/bin/env python3
"""Micro-benchmark of the tp_linear MFMA kernel on the GEMM shapes of the TokenPacker path at
B=256 (run on the GPU box).  Prints TFLOP/s per (shape, tile, xcd-swizzle) and, as a yard-stick,
torch.matmul (hipBLASLt/rocBLAS) on the same operands.  Random normal data (guide §5.4 rule 25)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import _capi  # noqa: E402

SHAPES = [  # name, M, N, K, flags
    ("kv_layer0", 256 * 576, 2048, 4096, _capi.TP_LINEAR_GELU),
    ("kv_layer2", 256 * 576, 1024, 1024, _capi.TP_LINEAR_ROW_STATS),
    ("q_side", 256 * 144, 1024, 1024, 0),
    ("mlp0", 256 * 144, 4096, 1024, _capi.TP_LINEAR_GELU),
    ("mlp2", 256 * 144, 4096, 4096, 0),
]


def time_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    lib = _capi.load_library()
    dtype = torch.bfloat16
    results = []
    for name, M, N, K, flags in SHAPES:
        A = torch.randn(M, K, device="cuda", dtype=torch.float32).to(dtype)
        W = (torch.randn(N, K, device="cuda", dtype=torch.float32) * K ** -0.5).to(dtype)
        bias = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda", dtype=dtype)
        stats = torch.empty(8 * M * 2, device="cuda")
        fl = 2.0 * M * N * K
        for tile in (128, 256):
            for swz in (1, 0):
                _capi.set_tuning(_capi.TP_TUNE_XCD_SWIZZLE, swz)
                args = _capi.tp_linear_args()
                args.M, args.N, args.K, args.dtype, args.flags = M, N, K, _capi.TP_BF16, flags
                args.lda, args.ldc, args.tile = K, N, tile
                args.A, args.W, args.C, args.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr()
                args.row_stats_out = stats.data_ptr()
                st = torch.cuda.current_stream().cuda_stream

                def run():
                    rc = lib.tp_linear(ctypes.byref(args), st)
                    assert rc == 0, _capi.last_error()
                ms = time_ms(run)
                results.append(dict(shape=name, M=M, N=N, K=K, tile=tile, swizzle=swz, ms=round(ms, 4),
                                    tflops=round(fl / ms / 1e9, 1)))
                print(results[-1], flush=True)
        _capi.set_tuning(_capi.TP_TUNE_XCD_SWIZZLE, 1)
        ms = time_ms(lambda: torch.matmul(A, W.t()))
        results.append(dict(shape=name, M=M, N=N, K=K, tile="torch.matmul", ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
        print(results[-1], flush=True)
        del A, W, C
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(results, open("gpurun_out/gemm_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
