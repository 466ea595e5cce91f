#!/usr/bin/env python3
"""The one-wave-per-SIMD GEMM kernel (tp_gemm4.hip, TP_TUNE_PAIR_GEMM = 3) against the ping-pong kernel on single launches:
bit-identity on a few shapes / epilogues, then HIP-event timings of the forward's large launches, arms interleaved.

    python tools/solo_ab.py [--out gpurun_out/solo_ab.json] [--debug D]   (D: TP_TUNE_PAIR_DEBUG probe build, timing only)
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenpacker_amd import _capi  # noqa: E402
from tests import gpu_util as gu  # noqa: E402

G, F, S = _capi.TP_LINEAR_GELU, _capi.TP_LINEAR_LN_FOLD, _capi.TP_LINEAR_ROW_STATS


def rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


def run(mode, fn):
    _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, mode)
    try:
        return fn()
    finally:
        _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)


def check():
    bad = 0
    for dtype in (torch.bfloat16, torch.float16):
        for (M, N, K) in [(256, 256, 128), (512, 512, 256), (1000, 1024, 1024), (300, 256, 4096), (77, 256, 1024), (18432, 2048, 4096)]:
            A = rand((M, K), dtype, 1)
            W = rand((N, K), dtype, 2, K ** -0.5)
            bias = rand((N,), torch.float32, 3)
            for out_dtype in (torch.float16, torch.bfloat16, torch.float32):
                for flags, b, st in ((0, None, False), (G, bias, False), (0, bias, out_dtype != torch.float32)):
                    fn = lambda: gu.linear(A, W, bias=b, flags=flags, out_dtype=out_dtype, want_stats=st, tile=0)  # noqa: E731
                    ref = run(1, fn)
                    got = run(3, fn)
                    again = run(3, fn)
                    if st:
                        ok = all(torch.equal(x, y) for x, y in zip(got, ref)) and all(torch.equal(x, y) for x, y in zip(got, again))
                    else:
                        ok = torch.equal(got, ref) and torch.equal(got, again)
                    if not ok:
                        bad += 1
                        g0 = got[0] if st else got
                        r0 = ref[0] if st else ref
                        print("MISMATCH", dtype, out_dtype, (M, N, K), flags, st, gu.describe_mismatch(g0, r0, "solo", 0.0)[:600], flush=True)
    print(f"bit-identity check: {bad} mismatches", flush=True)
    return bad


def time_shapes(out, debug):
    shapes = [("kv_layer0", 147456, 2048, 4096, torch.bfloat16, G), ("mlp2", 36864, 4096, 4096, torch.float16, 0),
              ("mlp0", 36864, 4096, 1024, torch.float16, G), ("k1024", 147456, 1024, 1024, torch.float16, 0),
              ("kv_layer0_B32", 18432, 2048, 4096, torch.bfloat16, G),
              ("probe_k4096", 36864, 4096, 4096, torch.float16, 0), ("probe_k1024", 36864, 4096, 1024, torch.float16, 0)]
    if debug:
        shapes = shapes[-2:]
    res = {}
    for name, M, N, K, dtype, flags in shapes:
        A = rand((M, K), dtype, 1)
        W = rand((N, K), dtype, 2, K ** -0.5)
        bias = rand((N,), torch.float32, 3)
        odt = torch.float16 if name != "mlp2" else torch.bfloat16      # (the probe builds exist for fp16 -> fp16, contiguous A)
        times = {1: [], 3: []}
        for mode in (1, 3):
            run(mode, lambda: gu.linear(A, W, bias=bias, flags=flags, out_dtype=odt))
        for _ in range(7):
            for mode in (1, 3):
                _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, mode)
                _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, debug if mode == 3 else 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    gu.linear(A, W, bias=bias, flags=flags, out_dtype=odt, sync=False)
                e1.record()
                torch.cuda.synchronize()
                times[mode].append(e0.elapsed_time(e1) / 5)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 0)
        flop = 2.0 * M * N * K
        r = {m: statistics.median(t) for m, t in times.items()}
        res[name] = {"M": M, "N": N, "K": K, "pingpong_ms": round(r[1], 4), "solo_ms": round(r[3], 4),
                     "pingpong_tflops": round(flop / r[1] / 1e9, 1), "solo_tflops": round(flop / r[3] / 1e9, 1),
                     "solo_over_pingpong": round(r[3] / r[1], 4)}
        print(name, res[name], flush=True)
    if "probe_k4096" in res:
        for k in ("pingpong_ms", "solo_ms"):
            t4, t1 = res["probe_k4096"][k], res["probe_k1024"][k]
            per_ktile = (t4 - t1) / 9 / 48 * 1e3
            print(f"{k}: {per_ktile:.3f} us per K-tile, {t1 / 9 * 1e3 - 16 * per_ktile:.2f} us fixed per tile (9 tiles per CU)", flush=True)
            res[k + "_fit"] = {"us_per_ktile": round(per_ktile, 4), "us_fixed_per_tile": round(t1 / 9 * 1e3 - 16 * per_ktile, 3)}
    if out:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump({"debug": debug, "shapes": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/solo_ab.json")
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    if not a.no_check and a.debug == 0:
        check()
    time_shapes(a.out, a.debug)
