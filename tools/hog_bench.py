#!/usr/bin/env python3
"""Persistent-GEMM tile scheduling under CU contention: time the B=256 forward while `--hog` CUs are kept busy by
another stream (tp_test_occupy_cus of libtokenpacker_exp.so — what a collective's kernels do when the all-gather of step i overlaps the
forward of step i+1), static striding vs per-XCD tile queues (TP_TUNE_DYNAMIC_TILES), and with r CUs per XCD RESERVED
for the other stream (TP_TUNE_RESERVE_CUS: the persistent GEMMs launch 256 - 8 r workgroups)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenpacker_amd import TokenPacker, _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hogs", type=int, nargs="+", default=[0, 16, 32, 64])
    ap.add_argument("--reserve", type=int, nargs="+", default=[0, 1, 2, 4])
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    B, dtype = args.batch, torch.bfloat16
    lib = _capi.load_library()
    m = TokenPacker(hidden_size=4096, scale_factor=2).to(device="cuda", dtype=dtype).eval().requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 576, 1024, generator=g, device="cuda").to(dtype)
    xm = torch.randn(B, 576, 4096, generator=g, device="cuda").to(dtype)
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    out = []
    with torch.no_grad():
        ref = m((x, xm))
        for hog in args.hogs:
            for dyn, rsv in [(0, 0)] + [(1, r) for r in args.reserve]:
                _capi.set_tuning(_capi.TP_TUNE_DYNAMIC_TILES, dyn)
                _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, rsv)
                ts = []
                for rep in range(6):
                    torch.cuda.synchronize()
                    if hog:
                        assert _capi.load_test_library().tp_test_occupy_cus(hog, 40000, sink.data_ptr(), side.cuda_stream) == 0   # ~40 ms
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        y = m((x, xm))
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 5)
                assert torch.equal(y, ref)
                rec = {"B": B, "hog_cus": hog, "dynamic_tiles": dyn, "reserve_cus_per_xcd": rsv,
                       "ms_per_forward": round(sorted(ts)[len(ts) // 2], 3)}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    _capi.set_tuning(_capi.TP_TUNE_DYNAMIC_TILES, 1)
    _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, 0)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/hog_bench_B{B}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
