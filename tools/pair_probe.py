#!/usr/bin/env python3
"""Timing probes of the pair GEMM kernel (TP_TUNE_PAIR_DEBUG): where a K-tile's time goes.  fp16 -> fp16 plain launches."""
import os as _os
_os.environ.setdefault("TP_LIB_VARIANT", "exp")     # the timing-probe instantiations live in libtokenpacker_exp.so only (make exp)
import ctypes, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import _capi

ARMS = [("pp_persistent", dict(PAIR_GEMM=1)), ("pair", dict(PAIR_GEMM=2)), ("pair_1wg", dict(PAIR_GEMM=2, PAIR_DEBUG=8)),
        ("no_dma", dict(PAIR_GEMM=2, PAIR_DEBUG=1)), ("no_reads", dict(PAIR_GEMM=2, PAIR_DEBUG=2)),
        ("no_barrier", dict(PAIR_GEMM=2, PAIR_DEBUG=3)), ("no_mfma", dict(PAIR_GEMM=2, PAIR_DEBUG=4)),
        ("no_dma_1wg", dict(PAIR_GEMM=2, PAIR_DEBUG=9)), ("no_reads_1wg", dict(PAIR_GEMM=2, PAIR_DEBUG=10)),
        ("no_mfma_1wg", dict(PAIR_GEMM=2, PAIR_DEBUG=12))]
SHAPES = [("K4096", 147456, 1024, 4096), ("K1024", 147456, 1024, 1024)]

def main():
    lib = _capi.load_library()
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        C = torch.empty(M, N, device="cuda", dtype=torch.float16); bias = torch.randn(N, device="cuda")
        a = _capi.tp_linear_args(); a.M, a.N, a.K, a.flags = M, N, K, 0
        a.dtype = a.out_dtype = _capi.TP_F16; a.lda, a.ldc, a.tile = K, N, 0
        a.A, a.W, a.C, a.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr()
        times = {n: [] for n, _ in ARMS}
        def run(sets):
            for k, v in _capi._TUNING_DEFAULTS.items(): _capi.set_tuning(k, v)
            for k, v in sets.items(): _capi.set_tuning(getattr(_capi, "TP_TUNE_" + k), v)
            assert lib.tp_linear(ctypes.byref(a), st) == 0, _capi.last_error()
        for n, sets in ARMS: run(sets)
        torch.cuda.synchronize()
        for _ in range(5):
            for n, sets in ARMS:
                run(sets)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): assert lib.tp_linear(ctypes.byref(a), st) == 0
                e1.record(); torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 5)
        for n, _ in ARMS:
            ms = statistics.median(times[n])
            # per K-tile and 256 x 256-tile equivalent on one CU: tiles256 = M/256 * N/256 over 256 CUs
            per_kt = ms * 1e3 / ((M / 256) * (N / 256) / 256) / (K / 64)
            rec = dict(shape=name, arm=n, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1), us_per_ktile_256sq=round(per_kt, 3))
            out.append(rec); print(rec, flush=True)
    for k, v in _capi._TUNING_DEFAULTS.items(): _capi.set_tuning(k, v)
    json.dump(out, open("gpurun_out/pair_probe.json", "w"), indent=1)

if __name__ == "__main__":
    main()
