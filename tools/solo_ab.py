#!/usr/bin/env python3
"""EXPERIMENT (round 5): the one-wave-per-SIMD 256 x 256 GEMM of round 4 (tokenpacker_amd/csrc/experimental/tp_gemm4.hip) with its
operands fetched by LDS-DMA (fetch 0, as measured in round 4) or staged through registers (fetch 1: buffer_load_dwordx4 -> VGPR ->
ds_write_b128), against the shipped ping-pong kernel — single launches through tp_linear's argument block, arms interleaved, HIP events.

    make -C tokenpacker_amd/csrc exp && python tools/solo_ab.py [--out gpurun_out/solo_ab.json]

Prints a bit-identity check first (the solo kernel shares the fragment layout, the MFMA order per accumulator and the epilogue), then
median times per shape and the per-K-tile / per-tile fit from the K = 1024 / 4096 pair of a 36864 x 4096 launch.
"""
import argparse
import ctypes
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


def rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


def make_args(A, W, bias, C, flags):
    a = _capi.tp_linear_args()
    a.M, a.N, a.K = A.shape[0], W.shape[0], W.shape[1]
    a.dtype, a.out_dtype, a.flags = DT[W.dtype], DT[C.dtype], flags
    a.lda, a.ldc = A.stride(0), W.shape[0]
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), C.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.tile = 256
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/solo_ab.json")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--dbg", type=int, default=0, help="TP_TUNE_PAIR_DEBUG probe build of the f16 -> f16 launches (1 no DMA | 2 no fragment reads | 4 no MFMAs; garbage results: skips the bit-identity check)")
    ap.add_argument("--vendor", action="store_true", help="add torch.matmul (hipBLASLt) on the probe shapes: its per-K-tile / per-tile fit")
    ap.add_argument("--tile-major", action="store_true", help="add the tile-major addressing probes of the solo kernel on the fp16 probe shapes (skips the bit-identity check)")
    ap.add_argument("--arms", default="", help="comma-separated subset of the solo arms (pingpong always runs)")
    a = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, "tokenpacker_amd", "libtokenpacker_exp.so"))
    lib.tp_linear.restype = ctypes.c_int
    lib.tp_linear.argtypes = [ctypes.POINTER(_capi.tp_linear_args), ctypes.c_void_p]
    lib.tp_exp_gemm4.restype = ctypes.c_int
    lib.tp_exp_gemm4.argtypes = [ctypes.POINTER(_capi.tp_linear_args), ctypes.c_void_p, ctypes.c_int]
    lib.tp_last_error.restype = ctypes.c_char_p
    stream = torch.cuda.current_stream().cuda_stream
    arms = {"pingpong": lambda args: lib.tp_linear(ctypes.byref(args), stream),
            "solo_dma": lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 0),
            "solo_reg": lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 1),
            # round 6: LDS-DMA with the SPREAD schedule (tp_gemm4.hip header: one memory instruction per MFMA slot at most, DMA pieces
            # over the whole K-tile behind a counted vmcnt, three barriers), and the same with odd waves one slot later
            "solo_spread": lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 2),
            "solo_spread_stagger": lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 3)}
    if a.tile_major:                        # timing probes (garbage results): the addresses of a tile-major operand layout
        arms["solo_tile_major"] = lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 4)
        arms["solo_tile_major_w"] = lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 5)
        arms["solo_tile_major_dma_nt"] = lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 6)     # (row-major; the DMA loads non-temporal — valid results)
        arms["solo_tile_major_dma_sc0"] = lambda args: lib.tp_exp_gemm4(ctypes.byref(args), stream, 7)    # (row-major; sc0)
    if a.arms:
        arms = {k: v for k, v in arms.items() if k == "pingpong" or k in a.arms.split(",") or k.startswith("solo_tile_major")}

    bad = 0
    if a.dbg:
        lib.tp_set_tuning.restype = ctypes.c_int
        lib.tp_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
        assert lib.tp_set_tuning(_capi.TP_TUNE_PAIR_DEBUG, a.dbg) == 0, lib.tp_last_error()
    for dtype in (() if (a.dbg or a.tile_major) else (torch.bfloat16, torch.float16)):
        for (M, N, K) in [(256, 256, 128), (512, 512, 256), (1000, 1024, 1024), (300, 256, 4096), (77, 256, 1024), (18432, 2048, 4096)]:
            A, W, bias = rand((M, K), dtype, 1), rand((N, K), dtype, 2, K ** -0.5), rand((N,), torch.float32, 3)
            for odt in (torch.float16, torch.bfloat16):
                for flags, b in ((0, None), (G, bias)):
                    outs = {}
                    for name, fn in arms.items():
                        C = torch.zeros(M, N, dtype=odt, device="cuda")
                        rc = fn(make_args(A, W, b, C, flags))
                        assert rc == 0, (name, lib.tp_last_error())
                        torch.cuda.synchronize()
                        outs[name] = C
                    for name in [k for k in arms if k != "pingpong"]:
                        if not torch.equal(outs[name], outs["pingpong"]):
                            bad += 1
                            d = (outs[name].float() - outs["pingpong"].float()).abs()
                            print("MISMATCH", name, dtype, odt, (M, N, K), flags, "max|d|", float(d.max()), "count", int((d > 0).sum()), flush=True)
    print(f"bit-identity check: {bad} mismatches", flush=True)

    shapes = [("kv_layer0", 147456, 2048, 4096, torch.bfloat16, torch.float16, G), ("mlp2", 36864, 4096, 4096, torch.float16, torch.bfloat16, 0),
              ("mlp0", 36864, 4096, 1024, torch.float16, torch.float16, G), ("k1024", 147456, 1024, 1024, torch.float16, torch.float16, 0),
              ("probe_k4096", 36864, 4096, 4096, torch.float16, torch.float16, 0), ("probe_k1024", 36864, 4096, 1024, torch.float16, torch.float16, 0)]
    res = {"mismatches": bad, "dbg": a.dbg}
    if a.dbg or a.tile_major:
        shapes = shapes[-2:]
    if a.vendor:
        arms = dict(arms)
        arms["vendor_matmul"] = None
    for name, M, N, K, dt, odt, flags in shapes:
        A, W, bias = rand((M, K), dt, 1), rand((N, K), dt, 2, K ** -0.5), rand((N,), torch.float32, 3)
        C = torch.empty(M, N, dtype=odt, device="cuda")
        args = make_args(A, W, bias, C, flags)
        if a.vendor:
            Wt, Cv = W.t(), torch.empty(M, N, dtype=dt, device="cuda")
            arms["vendor_matmul"] = lambda args, A=A, Wt=Wt, Cv=Cv: torch.matmul(A, Wt, out=Cv)
        for fn in arms.values():
            fn(args)
        torch.cuda.synchronize()
        times = {k: [] for k in arms}
        for _ in range(a.rounds):
            for k, fn in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    fn(args)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 5)
        r = {k: round(statistics.median(v), 4) for k, v in times.items()}
        res[name] = {"M": M, "N": N, "K": K, "ms": r, **{k + "_over_pingpong": round(r[k] / r["pingpong"], 4) for k in r if k != "pingpong"}}
        print(name, res[name], flush=True)
    for k in arms:
        t4, t1 = res["probe_k4096"]["ms"][k], res["probe_k1024"]["ms"][k]
        per = (t4 - t1) / 9 / 48 * 1e3
        res[k + "_fit"] = {"us_per_ktile": round(per, 4), "us_fixed_per_tile": round(t1 / 9 * 1e3 - 16 * per, 3)}
        print(k, res[k + "_fit"], flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
