#!/usr/bin/env python3
"""The parity claim as a DISTRIBUTION: seeds x scale_factor in {2, 3, 4} x {bf16 model with fp32 output, fp16
model}, HIP path against the fp64 oracle on the SAME rounded operands, metric max|y - y_ref| / max|y_ref| and
rel-L2.  Prints one line per configuration and writes the per-seed values + summary as JSON.

    python tools/parity_sweep.py [--seeds 128] [--out gpurun_out/parity_seed_sweep.json]
"""
import argparse
import itertools
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tokenpacker_oracle as orc            # (the checker: this tool is test infrastructure, like tests/)
from tokenpacker_amd import TokenPacker, synth


def sweep(seeds, D=256, B=4, scale_factors=(2, 3, 4), log=print, tags=("bf16_fp32out", "fp16")):
    summary = {}
    for s, (dtype, tag) in itertools.product(scale_factors, ((torch.bfloat16, "bf16_fp32out"), (torch.float16, "fp16"))):
        if tag not in tags:
            continue
        errs, l2s = [], []
        for seed in range(seeds):
            params = synth.make_params(9000 + 17 * seed + s, D)
            x, xm = synth.make_inputs(9500 + 31 * seed + s, B, dtype)
            m = TokenPacker(hidden_size=D, scale_factor=s)
            m.load_state_dict(params, strict=True)
            m = m.to(device="cuda", dtype=dtype).eval().requires_grad_(False)
            m.output_fp32 = dtype == torch.bfloat16
            with torch.no_grad():
                y = m((x.cuda(), xm.cuda()))
            p_lp = {k: v.to(dtype) for k, v in params.items()}
            y_exact = orc.forward(p_lp, x, xm, scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype)
            errs.append(orc.rel_err(y, y_exact))
            l2s.append(orc.rel_l2(y, y_exact))
        key = f"s{s}_{tag}"
        q = sorted(errs)
        summary[key] = {"seeds": seeds, "median": statistics.median(errs), "p90": q[(len(q) * 9) // 10], "max": max(errs), "min": min(errs),
                        "l2_median": statistics.median(l2s), "l2_max": max(l2s), "rel_max_per_seed": [round(e, 7) for e in errs]}
        r = summary[key]
        log(f"[parity-sweep] {key}: {seeds} seeds, rel-max median {r['median']:.3e} p90 {r['p90']:.3e} max {r['max']:.3e} min {r['min']:.3e}"
            f" | rel-L2 median {r['l2_median']:.3e} max {r['l2_max']:.3e}")
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=128)
    ap.add_argument("--out", default="gpurun_out/parity_seed_sweep.json")
    ap.add_argument("--scale-factors", type=int, nargs="+", default=[2, 3, 4])
    ap.add_argument("--tags", nargs="+", default=["bf16_fp32out", "fp16"])
    args = ap.parse_args()
    summary = sweep(args.seeds, scale_factors=tuple(args.scale_factors), tags=tuple(args.tags))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(summary, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
