#!/usr/bin/env python3
"""Training-step timing of the projector on one MI355X: HIP forward+backward (tp_forward_train / tp_backward
through the autograd node) next to the reference's op sequence under PyTorch-ROCm eager autograd
(oracle/reference_ops.py).  Batch 32 is the reference's per-GPU pretraining batch (scripts/v1_5/pretrain.sh:19).
A step is forward + backward + OPTIMIZER STEP (plain SGD on the 23 parameters, `--optimizer none` to leave it out):
the weight update is what makes the next forward re-pack the kernel-side weight image (casts, LayerNorm folds) and
the backward re-transpose the weights, so it belongs inside the timed step.  FLOPs are priced on the work the
backward actually has to do: no input gradient for k/v_proj_1[0] (the CLIP features come from a frozen tower).

    python tools/train_bench.py [--batches 32 256] [--scale-factor 2] [--out gpurun_out/train_bench.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.reference_ops import eager_forward  # noqa: E402  (baseline only)
from tokenpacker_amd import TokenPacker  # noqa: E402


def flops_fwd(B, s, D):
    N, M, E = 576, (24 // s) ** 2, 1024
    return 2.0 * B * (N * 4096 * E * 2 + N * E * E * 4 + M * E * E * 3 + M * E * D + M * D * D)


def flops_train(B, s, D):
    """forward + weight gradients of every linear + input gradients of every linear EXCEPT the two first K/V layers
    (their input is x_multi, which gets no gradient: 2 x 4.83 GF / image of dgrad that nobody has to compute)."""
    N, E = 576, 1024
    return 3.0 * flops_fwd(B, s, D) - 2.0 * B * (N * 4096 * E * 2)


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[32, 256])
    ap.add_argument("--scale-factor", type=int, default=2)
    ap.add_argument("--hidden-size", type=int, default=4096)
    ap.add_argument("--out", default="gpurun_out/train_bench.json")
    ap.add_argument("--hip-only", action="store_true", help="skip the eager baseline (profiling runs)")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "none"])
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="library tuning knob, e.g. BWD_CHAIN=1")
    args = ap.parse_args()
    from tokenpacker_amd import _capi
    for kv in args.tune:
        key, val = kv.split("=")
        _capi.set_tuning(getattr(_capi, "TP_TUNE_" + key.upper()), int(val))
    s, D, dtype = args.scale_factor, args.hidden_size, torch.bfloat16
    results = []
    for B in args.batches:
        torch.manual_seed(0)
        m = TokenPacker(hidden_size=D, scale_factor=s).to(device="cuda", dtype=dtype)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(B, 576, 1024, generator=g, device="cuda").to(dtype)
        xm = torch.randn(B, 576, 4096, generator=g, device="cuda").to(dtype)
        w = torch.randn(B, (24 // s) ** 2, D, generator=g, device="cuda").to(dtype)

        import copy
        m_eager = copy.deepcopy(m)                       # each pipeline trains its own copy of the weights
        opt = torch.optim.SGD(m.parameters(), lr=1e-7) if args.optimizer == "sgd" else None
        opt_e = torch.optim.SGD(m_eager.parameters(), lr=1e-7) if args.optimizer == "sgd" else None

        def step_hip():
            m.zero_grad(set_to_none=True)
            (m((x, xm)) * w).sum().backward()
            if opt is not None:
                opt.step()

        def step_eager():
            m_eager.zero_grad(set_to_none=True)
            (eager_forward(m_eager, x, xm) * w).sum().backward()
            if opt_e is not None:
                opt_e.step()

        def fwd_hip():
            with torch.no_grad():
                m((x, xm))

        iters = 20 if B <= 64 else 8
        ms_hip = timed(step_hip, iters)
        ms_eager = timed(step_eager, iters) if not args.hip_only else float("nan")
        ms_fwd = timed(fwd_hip, iters)
        # gradient agreement of the two bf16 pipelines (sanity, loose)
        # (ln_k_1.bias and the k third of in_proj_bias have a mathematically zero gradient: round-off in both)
        E = 1024
        if args.hip_only:
            print(json.dumps({"B": B, "hip_fwd_bwd_ms": round(ms_hip, 3), "hip_fwd_only_ms": round(ms_fwd, 3)}), flush=True)
            continue
        finite = {"hip": all(bool(torch.isfinite(p.float()).all()) for p in m.parameters()),
                  "eager": all(bool(torch.isfinite(p.float()).all()) for p in m_eager.parameters())}
        m_eager.load_state_dict(m.state_dict())          # gradient agreement on identical weights
        step_eager(); ge = {k: p.grad.float().clone() for k, p in m_eager.named_parameters()}
        m.load_state_dict(m_eager.state_dict()) if False else None
        m.zero_grad(set_to_none=True); (m((x, xm)) * w).sum().backward()
        gh = {k: p.grad.float().clone() for k, p in m.named_parameters()}
        for d in (ge, gh):
            d.pop("ln_k_1.bias")
            d["clip_attn.in_proj_bias"] = torch.cat([d["clip_attn.in_proj_bias"][:E], d["clip_attn.in_proj_bias"][2 * E:]])
        agree = max(float((gh[k] - ge[k]).norm() / (ge[k].norm() + 1e-20)) for k in ge)
        rec = {"B": B, "scale_factor": s, "D": D, "dtype": "bf16", "hip_fwd_bwd_ms": round(ms_hip, 3),
               "hip_fwd_only_ms": round(ms_fwd, 3), "eager_rocm_fwd_bwd_ms": round(ms_eager, 3),
               "speedup": round(ms_eager / ms_hip, 3), "images_per_s_train": round(B / ms_hip * 1e3, 1),
               "optimizer_in_step": args.optimizer,
               "train_tflops_algorithmic": round(flops_train(B, s, D) / ms_hip / 1e9, 1),
               "train_gflop_per_image": round(flops_train(1, s, D) / 1e9, 2),
               "weights_finite_after_training": finite, "max_param_grad_rel_l2_vs_eager": agree, "tune": args.tune}
        print(json.dumps(rec), flush=True)
        results.append(rec)
        del m, m_eager, x, xm, w
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
