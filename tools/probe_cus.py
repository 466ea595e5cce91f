import os as _os
_os.environ.setdefault("TP_LIB_VARIANT", "exp")     # the timing-probe instantiations live in libtokenpacker_exp.so only (make exp)
import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
from tokenpacker_amd import _capi
from tests import gpu_util as gu
def rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)
M, N = 36864, 4096
ops = {K: (rand((M, K), torch.float16, 1), rand((N, K), torch.float16, 2, K ** -0.5), rand((N,), torch.float32, 3)) for K in (1024, 4096)}
_capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 1)
for probe in (79, 0, 64):
    for r in (0, 4, 7):
        _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, r)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, probe << 4)
        t = {}
        for K, (A, W, b) in ops.items():
            gu.linear(A, W, bias=b, out_dtype=torch.float16, tile=256)
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(5): gu.linear(A, W, bias=b, out_dtype=torch.float16, tile=256, sync=False)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
            t[K] = statistics.median(ts)
        ncu = (32 - r) * 8
        per = (t[4096] - t[1024]) * ncu / (2304 * 48) * 1e3
        print(f"probe {probe} reserve {r} ({ncu} CUs): K1024 {t[1024]:.4f} K4096 {t[4096]:.4f} ms -> {per:.3f} us per K-tile and CU", flush=True)
_capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, 0); _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 0)
