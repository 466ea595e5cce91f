#!/usr/bin/env python3
"""Parameter-gradient parity as a DISTRIBUTION (round 6): seeds x scale_factor in {2, 3, 4} x gradient chain {fp16 behind a dynamic
power-of-two scale (TP_TUNE_BWD_CHAIN = 0, default), bf16 (= 1)} for a bf16 model, and the fp16 model, HIP backward against fp64
autograd on the oracle with the same rounded operands.  Metric per parameter: rel-L2 of the gradient error against
max(rms(g_ref), 0.1 * largest rms among same-shaped parameters) — tests/test_gpu_backward.py's; reported: the worst parameter per seed.

    python tools/grad_parity_sweep.py [--seeds 32] [--out gpurun_out/grad_parity_sweep.json]
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tokenpacker_oracle as orc            # (the checker: this tool is test infrastructure, like tests/)
from tokenpacker_amd import TokenPacker, _capi, synth


def worst_param_err(dtype, chain, s, D, B, seed):
    params = synth.make_params(7000 + 13 * seed + s, D)
    x, xm = synth.make_inputs(7500 + 29 * seed + s, B, dtype)
    w = torch.randn(B, (24 // s) ** 2, D, generator=torch.Generator().manual_seed(8000 + seed)).to(dtype)
    m = TokenPacker(hidden_size=D, scale_factor=s)
    m.load_state_dict(params)
    m = m.to(device="cuda", dtype=dtype).train()
    m.tuning = _capi.TuningContext(bwd_chain=chain)
    m((x.cuda(), xm.cuda())).backward(w.cuda())
    torch.cuda.synchronize()
    got = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}
    p_lp = {k: v.to(dtype) for k, v in params.items()}
    ref_p = {k: v.double().requires_grad_(True) for k, v in p_lp.items()}
    orc.forward(ref_p, x, xm, scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype).backward(w.double())
    want = {k: v.grad for k, v in ref_p.items()}
    rms = {k: float(v.norm()) / v.numel() ** 0.5 for k, v in want.items()}
    worst = 0.0
    for k in want:
        scale = max(rms[k], 0.1 * max(rms[j] for j in want if want[j].shape == want[k].shape))
        worst = max(worst, float((got[k] - want[k]).norm()) / want[k].numel() ** 0.5 / scale)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=32)
    ap.add_argument("--hidden-size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--out", default="gpurun_out/grad_parity_sweep.json")
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    out = {}
    for s in (2, 3, 4):
        for tag, dtype, chain in (("bf16_model_fp16_chain", torch.bfloat16, 0), ("bf16_model_bf16_chain", torch.bfloat16, 1), ("fp16_model", torch.float16, 0)):
            errs = [worst_param_err(dtype, chain, s, a.hidden_size, a.batch, seed) for seed in range(a.seeds)]
            q = sorted(errs)
            out[f"s{s}_{tag}"] = {"seeds": a.seeds, "median": statistics.median(errs), "p90": q[(len(q) * 9) // 10], "max": max(errs), "per_seed": [round(e, 6) for e in errs]}
            r = out[f"s{s}_{tag}"]
            print(f"[grad-sweep] s{s}_{tag}: {a.seeds} seeds, worst-parameter rel-L2 median {r['median']:.3e} p90 {r['p90']:.3e} max {r['max']:.3e}", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
