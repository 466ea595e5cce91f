#!/usr/bin/env python3
"""What bounds the fixed cost per tile of the persistent GEMM (round 6)?  Fit time = tiles_per_CU * (nk * t_ktile + t_fixed) from a K = 1024 /
K = 4096 pair, with all 32 CUs of an XCD active and with 25 (TP_TUNE_RESERVE_CUS = 7; M chosen so that both launches are whole rounds): a
per-CU bound (store issue, epilogue arithmetic) leaves t_fixed unchanged, a per-XCD / HBM write bound scales it with the active CUs.  Also:
the write-only and copy rates of the box (torch), and t_fixed by output width (f16 vs f32 output = twice the bytes per store instruction).

    python tools/probes/store_bound_probe.py [--out gpurun_out/store_bound.json]
"""
import argparse, ctypes, json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from tokenpacker_amd import _capi  # noqa: E402
import solo_ab  # noqa: E402


def timeit(fn, rounds=5, inner=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default="gpurun_out/store_bound.json"); a = ap.parse_args()
    lib = _capi.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    x = torch.empty(302 * 2 ** 20 // 2, dtype=torch.float16, device="cuda"); y = torch.empty_like(x)
    t = timeit(lambda: x.zero_()); res["write_only_TBps"] = round(x.numel() * 2 / t / 1e9, 3)
    t = timeit(lambda: y.copy_(x)); res["copy_TBps_read_plus_write"] = round(2 * x.numel() * 2 / t / 1e9, 3)
    print(res, flush=True)
    # (a) whole rounds on 256 / 224 / 200 CUs (persistent launches), (b) ONE tile per workgroup on 256 / 128 / 64 / 32 CUs (xcd_remap spreads
    # them over the XCDs; t_fixed then includes the launch overhead, the same for every row).  Cases interleaved, 3 repetitions.
    cases = [("persist", 0, 36864), ("persist", 7, 38400), ("persist", 16, 36864), ("persist", 24, 18432), ("persist", 28, 9216)]
    bufs = {}
    for kind, reserve, M in cases:
        for odt in (torch.float16, torch.float32):
            for K in (1024, 4096):
                A, W = solo_ab.rand((M, K), torch.float16, 1), solo_ab.rand((4096, K), torch.float16, 2, K ** -0.5)
                C = torch.empty(M, 4096, dtype=odt, device="cuda")
                bufs[(kind, reserve, M, odt, K)] = (A, W, C, solo_ab.make_args(A, W, None, C, 0))
    times = {k: [] for k in bufs}
    for rep in range(4):
        for key, (A, W, C, args) in bufs.items():
            assert lib.tp_set_tuning(_capi.TP_TUNE_RESERVE_CUS, key[1]) == 0
            fn = lambda: lib.tp_linear(ctypes.byref(args), stream)
            for _ in range(10):
                assert fn() == 0, lib.tp_last_error()
            t = timeit(fn, rounds=3, inner=10)
            if rep:
                times[key].append(t)
    for kind, reserve, M in cases:
        cus = (32 - reserve) * 8
        tiles = (M // 256) * 16
        per_cu = tiles / cus if kind == "persist" else 1.0
        for odt in (torch.float16, torch.float32):
            ms = {K: statistics.median(times[(kind, reserve, M, odt, K)]) for K in (1024, 4096)}
            per = (ms[4096] - ms[1024]) / per_cu / 48 * 1e3
            fixed = ms[1024] / per_cu * 1e3 - 16 * per
            key = f"{kind}_cus{cus if kind == 'persist' else tiles}_{str(odt).split('.')[-1]}"
            res[key] = {"M": M, "tiles_per_cu": per_cu, "ms_k1024": round(ms[1024], 4), "ms_k4096": round(ms[4096], 4), "us_per_ktile": round(per, 4), "us_fixed_per_tile": round(fixed, 3)}
            print(key, res[key], flush=True)
    lib.tp_set_tuning(_capi.TP_TUNE_RESERVE_CUS, 0)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
