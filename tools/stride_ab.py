#!/usr/bin/env python3
"""Row strides of the GEMM operands: power of two against + 64 elements (round 5).

tools/probes/operand_fetch_probe.hip: an LDS-DMA instruction whose 8 rows are 8 KiB apart fetches at 28 GB/s per CU, the same rows
8 KiB + 128 B apart at 123 GB/s (1 KiB contiguous: 137) — with K = 1024 / 4096 contiguous fp16 / bf16 operands EVERY row of a K-tile
lies in the same L2 channel class.  This tool times the path's GEMM shapes through tp_linear with each of A, W and C either contiguous
(row stride K resp. N elements) or padded by 64 elements, arms interleaved, and checks that the results are the same bits.

    python tools/stride_ab.py [--out gpurun_out/stride_ab.json]
"""
import argparse
import ctypes
import itertools
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenpacker_amd import _capi  # noqa: E402

DT = {torch.bfloat16: _capi.TP_BF16, torch.float16: _capi.TP_F16, torch.float32: _capi.TP_F32}
G = _capi.TP_LINEAR_GELU
PAD = 64


def rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


def strided(t, pad):
    """The same values in a buffer whose rows are `pad` elements further apart."""
    if not pad:
        return t.contiguous()
    buf = torch.zeros(t.shape[0], t.shape[1] + pad, dtype=t.dtype, device=t.device)
    buf[:, :t.shape[1]] = t
    return buf[:, :t.shape[1]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/stride_ab.json")
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    lib = _capi.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    shapes = [("kv_layer0", 147456, 2048, 4096, torch.bfloat16, torch.float16, G), ("mlp2", 36864, 4096, 4096, torch.float16, torch.bfloat16, 0),
              ("mlp0", 36864, 4096, 1024, torch.float16, torch.float16, G), ("k1024", 147456, 1024, 1024, torch.float16, torch.float16, 0),
              ("mlp2_B32", 4608, 4096, 4096, torch.float16, torch.bfloat16, 0)]
    arms = [(0, 0, 0), (PAD, 0, 0), (0, PAD, 0), (0, 0, PAD), (PAD, PAD, 0), (PAD, PAD, PAD)]
    res = {}
    for name, M, N, K, dt, odt, flags in shapes:
        A0, W0, bias = rand((M, K), dt, 1), rand((N, K), dt, 2, K ** -0.5), rand((N,), torch.float32, 3)
        ops = {}
        for pa, pw, pc in arms:
            A, W = strided(A0, pa), strided(W0, pw)
            Cbuf = torch.empty(M, N + pc, dtype=odt, device="cuda")
            args = _capi.tp_linear_args()
            args.M, args.N, args.K = M, N, K
            args.dtype, args.out_dtype, args.flags = DT[dt], DT[odt], flags
            args.lda, args.ldc, args.ldw = A.stride(0), N + pc, (W.stride(0) if pw else 0)
            args.A, args.W, args.C, args.bias = A.data_ptr(), W.data_ptr(), Cbuf.data_ptr(), bias.data_ptr()
            args.tile = 0
            ops[(pa, pw, pc)] = (args, A, W, Cbuf)
        del A0, W0
        outs = {}
        for k, (args, _, _, Cbuf) in ops.items():
            rc = lib.tp_linear(ctypes.byref(args), stream)
            assert rc == 0, _capi.last_error()
            torch.cuda.synchronize()
            outs[k] = Cbuf[:, :N]
        same = all(torch.equal(outs[arms[0]], o) for o in outs.values())
        times = {k: [] for k in ops}
        for _ in range(a.rounds):
            for k, (args, _, _, _) in ops.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    lib.tp_linear(ctypes.byref(args), stream)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 5)
        base = statistics.median(times[arms[0]])
        res[name] = {"M": M, "N": N, "K": K, "bit_identical": bool(same),
                     "ms": {f"padA{pa}_padW{pw}_padC{pc}": round(statistics.median(times[(pa, pw, pc)]), 4) for pa, pw, pc in arms},
                     "vs_contiguous": {f"padA{pa}_padW{pw}_padC{pc}": round(statistics.median(times[(pa, pw, pc)]) / base, 4) for pa, pw, pc in arms},
                     "tflops_contiguous": round(2.0 * M * N * K / base / 1e9, 1),
                     "tflops_all_padded": round(2.0 * M * N * K / statistics.median(times[arms[-1]]) / 1e9, 1)}
        print(name, json.dumps(res[name]), flush=True)
        del ops, outs
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
