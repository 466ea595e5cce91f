#!/usr/bin/env python3
"""A handful of launches of one plain GEMM (36864 x 4096 x K fp16, or its whole-round equivalent on fewer CUs) for a counter pass:
    rocprofv3 --kernel-trace --pmc <counters> -- python tools/probes/gemm_once.py --reserve 24 --k 4096
"""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from tokenpacker_amd import _capi  # noqa: E402
import solo_ab  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--reserve", type=int, default=0); ap.add_argument("--k", type=int, default=4096); ap.add_argument("--n", type=int, default=12)
a = ap.parse_args()
lib = _capi.load_library()
M = 36864 * (32 - a.reserve) // 32
A, W = solo_ab.rand((M, a.k), torch.float16, 1), solo_ab.rand((4096, a.k), torch.float16, 2, a.k ** -0.5)
C = torch.empty(M, 4096, dtype=torch.float16, device="cuda")
args = solo_ab.make_args(A, W, None, C, 0)
assert lib.tp_set_tuning(_capi.TP_TUNE_RESERVE_CUS, a.reserve) == 0
stream = torch.cuda.current_stream().cuda_stream
for _ in range(a.n):
    assert lib.tp_linear(ctypes.byref(args), stream) == 0
torch.cuda.synchronize()
