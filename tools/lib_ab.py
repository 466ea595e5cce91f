#!/usr/bin/env python3
"""Same-process A/B of two BUILDS of libtokenpacker_hip.so on single tp_linear launches (arms interleaved, HIP events).

    python tools/lib_ab.py --old tokenpacker_amd/libtokenpacker_old.so [--out gpurun_out/lib_ab.json]

The new build is the in-tree library; both are dlopen'ed side by side (separate static state, one HIP runtime).  Also checks that
the two builds produce the same bits on every shape.
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


def open_lib(path):
    lib = ctypes.CDLL(path)
    lib.tp_linear.restype = ctypes.c_int
    lib.tp_linear.argtypes = [ctypes.POINTER(_capi.tp_linear_args), ctypes.c_void_p]
    lib.tp_last_error.restype = ctypes.c_char_p
    return lib


def make_args(A, W, bias, C, flags, tile):
    a = _capi.tp_linear_args()
    a.M, a.N, a.K = A.shape[0], W.shape[0], W.shape[1]
    a.dtype, a.out_dtype, a.flags = DT[W.dtype], DT[C.dtype], flags
    a.lda, a.ldc = A.stride(0), W.shape[0]
    a.A, a.W, a.C = A.data_ptr(), W.data_ptr(), C.data_ptr()
    a.bias = bias.data_ptr()
    a.tile = tile
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--old", required=True)
    ap.add_argument("--new", default=_capi.LIB_PATH, help="the other build (default: the in-tree library)")
    ap.add_argument("--out", default="gpurun_out/lib_ab.json")
    ap.add_argument("--rounds", type=int, default=9)
    a = ap.parse_args()
    libs = {"old": open_lib(os.path.abspath(a.old)), "new": open_lib(os.path.abspath(a.new))}
    stream = torch.cuda.current_stream().cuda_stream
    shapes = [("kv_layer0", 147456, 2048, 4096, torch.bfloat16, torch.float16, G), ("mlp2", 36864, 4096, 4096, torch.float16, torch.bfloat16, 0),
              ("mlp0", 36864, 4096, 1024, torch.float16, torch.float16, G), ("k1024", 147456, 1024, 1024, torch.float16, torch.float16, 0),
              ("kv_layer0_B32", 18432, 2048, 4096, torch.bfloat16, torch.float16, G),
              ("probe_k4096", 36864, 4096, 4096, torch.float16, torch.float16, 0), ("probe_k1024", 36864, 4096, 1024, torch.float16, torch.float16, 0)]
    res = {}
    for name, M, N, K, dt, odt, flags in shapes:
        A, W, bias = rand((M, K), dt, 1), rand((N, K), dt, 2, K ** -0.5), rand((N,), torch.float32, 3)
        outs = {k: torch.empty(M, N, dtype=odt, device="cuda") for k in libs}
        args = {k: make_args(A, W, bias, outs[k], flags, 256) for k in libs}
        for k, lib in libs.items():
            rc = lib.tp_linear(ctypes.byref(args[k]), stream)
            assert rc == 0, lib.tp_last_error()
        torch.cuda.synchronize()
        same = bool(torch.equal(outs["old"], outs["new"]))
        times = {k: [] for k in libs}
        for _ in range(a.rounds):
            for k, lib in libs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    lib.tp_linear(ctypes.byref(args[k]), stream)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 5)
        r = {k: statistics.median(v) for k, v in times.items()}
        res[name] = {"M": M, "N": N, "K": K, "old_ms": round(r["old"], 4), "new_ms": round(r["new"], 4),
                     "new_over_old": round(r["new"] / r["old"], 4), "bit_identical": same,
                     "new_tflops": round(2.0 * M * N * K / r["new"] / 1e9, 1)}
        print(name, res[name], flush=True)
    for k in ("old_ms", "new_ms"):
        t4, t1 = res["probe_k4096"][k], res["probe_k1024"][k]
        per = (t4 - t1) / 9 / 48 * 1e3
        res[k + "_fit"] = {"us_per_ktile": round(per, 4), "us_fixed_per_tile": round(t1 / 9 * 1e3 - 16 * per, 3)}
        print(k, res[k + "_fit"], flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
