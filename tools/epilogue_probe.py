#!/usr/bin/env python3
"""Per-tile overhead probe: the GEMM kernels at K = 64 (one K-tile) on the path's M, N — what remains is the
launch / prologue / epilogue cost per 256x256 tile for each epilogue flavour (run on the GPU box)."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import _capi  # noqa: E402

G, S, F = _capi.TP_LINEAR_GELU, _capi.TP_LINEAR_ROW_STATS, _capi.TP_LINEAR_LN_FOLD


def main():
    lib = _capi.load_library()
    st = torch.cuda.current_stream().cuda_stream
    M, N = 147456, 1024
    for K in (64, 128):
        A = torch.randn(M, K, device="cuda").to(torch.float16)
        W = torch.randn(N, K, device="cuda").to(torch.float16)
        bias = torch.randn(N, device="cuda")
        colsum = torch.randn(N, device="cuda")
        mr = torch.rand(M, 2, device="cuda") + 0.5
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
        stats = torch.empty(8 * M * 2, device="cuda")
        for fname, flags in (("plain", 0), ("stats", S), ("lnfold", F), ("gelu", G)):
            for vname, tile_knob in (("full_tiles", 0), ("half_tiles", 2)):
                a = _capi.tp_linear_args()
                a.M, a.N, a.K, a.flags = M, N, K, flags
                a.dtype, a.out_dtype = _capi.TP_F16, _capi.TP_F16
                a.lda, a.ldc, a.tile = K, N, (256 if tile_knob == 0 else 0)
                a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr()
                a.row_stats_out, a.row_mean_rstd, a.colsum = stats.data_ptr(), mr.data_ptr(), colsum.data_ptr()
                _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, tile_knob)
                if tile_knob == 2 and K < 128:
                    continue
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
                tiles = (M // 256) * (N // 256)
                print(f"K={K:4d} {fname:7s} {vname:10s} {ms * 1e3:8.1f} us  = {ms * 1e3 / (tiles / 256):6.2f} us per tile-round "
                      f"(write {M * N * 2 / ms / 1e6:7.1f} GB/s)", flush=True)
    _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, 0)


if __name__ == "__main__":
    main()
