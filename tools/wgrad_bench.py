#!/usr/bin/env python3
"""Weight-gradient contraction dW = dY^T · X at the backward's shapes (B = 256): tp_wgrad (K-major operands, no transposed
copies) next to torch.matmul(dy.t(), x) (hipBLASLt TN) on the same box."""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import _capi  # noqa: E402

SHAPES = [("kv_layer0", 147456, 2048, 4096), ("kv_layer2", 147456, 1024, 1024), ("kv_inproj", 147456, 1024, 1024),
          ("mlp2", 36864, 4096, 4096), ("mlp0", 36864, 4096, 1024), ("q_side", 36864, 1024, 1024)]


def timed(fn, iters=10):
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts[1:])


def main():
    lib = _capi.load_library()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for name, R, N, K in SHAPES:
        dy = torch.randn(R, N, device="cuda").to(torch.bfloat16)
        x = torch.randn(R, K, device="cuda").to(torch.bfloat16)
        dw = torch.empty(N, K, dtype=torch.bfloat16, device="cuda")
        ws = torch.empty(lib.tp_wgrad_workspace_bytes(N, K), dtype=torch.uint8, device="cuda")

        def ours():
            assert lib.tp_wgrad(dy.data_ptr(), N, x.data_ptr(), K, 0, 0, R, N, K, _capi.TP_BF16, dw.data_ptr(), _capi.TP_BF16,
                                0, ws.data_ptr(), ws.numel(), st) == 0

        rpad = (R + 1023) // 1024 * 1024
        xt = torch.zeros(K, rpad, dtype=torch.bfloat16, device="cuda")
        xt[:, :R] = x.t()

        def ours_tn():
            assert lib.tp_wgrad(dy.data_ptr(), N, xt.data_ptr(), rpad, 0, 0, R, N, K, _capi.TP_BF16, dw.data_ptr(), _capi.TP_BF16,
                                _capi.TP_WGRAD_X_TRANSPOSED, ws.data_ptr(), ws.numel(), st) == 0

        ms = timed(ours)
        ms_tn = timed(ours_tn)
        ms_t = timed(lambda: torch.matmul(dy.t(), x))
        fl = 2.0 * R * N * K
        row = {"shape": name, "rows": R, "n_out": N, "k_in": K, "tp_wgrad_ms": round(ms, 4), "tp_wgrad_tflops": round(fl / ms / 1e9, 1),
               "tp_wgrad_xT_ms": round(ms_tn, 4), "torch_tn_ms": round(ms_t, 4), "torch_tn_tflops": round(fl / ms_t / 1e9, 1)}
        print(json.dumps(row), flush=True)
        out.append(row)
        del dy, x, dw, ws, xt
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
