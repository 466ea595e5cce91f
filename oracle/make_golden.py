#!/usr/bin/env python3
"""Generate ``tests/golden/*.npz`` by running the REAL reference module on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python oracle/make_golden.py

The reference class is imported by FILE PATH (``import llava`` fails under the installed
transformers) from
``/root/reference/llava/model/multimodal_projector/builder.py`` — nothing is copied from it.
Parameters and inputs come from ``tokenpacker_amd.synth`` (seeded, CPU) and are loaded into the
reference through its own ``load_state_dict`` (which also pins the state-dict contract,
SURVEY.md §8b).  Inputs are NOT stored (they are regenerated from the seed); their sha256 digest
is, so RNG drift is detected instead of silently mis-compared.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenpacker_amd import synth  # noqa: E402

from oracle import reference_loader  # noqa: E402
from oracle import tokenpacker_oracle as orc  # noqa: E402

REF_FILE = reference_loader.BUILDER_FILE
OUT_DIR = os.path.join(ROOT, "tests", "golden")

# name, scale_factor, hidden_size, B, param seed, input seed, row stride of the stored output
CASES = [
    ("s2_D256_B2", 2, 256, 2, 11, 101, 1),
    ("s3_D256_B2", 3, 256, 2, 12, 102, 1),
    ("s4_D256_B2", 4, 256, 2, 13, 103, 1),
    ("s2_D4096_B1", 2, 4096, 1, 14, 104, 2),
    ("s3_D5120_B1", 3, 5120, 1, 15, 105, 1),
]
INTER_STRIDE = 7     # intermediates are stored for rows ::7 only


def load_reference():
    return reference_loader.load_builder()


def run_case(ref, name, s, D, B, pseed, iseed, ostride):
    params = synth.make_params(pseed, D)
    x, xm = synth.make_inputs(iseed, B)

    cfg = type("Cfg", (), {"hidden_size": D, "scale_factor": s})()
    module = ref.build_vision_projector(cfg)          # the reference's own factory
    missing = module.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    module.eval()

    captured = {}

    def grab(key, index=None):
        def hook(_m, _inp, out):
            captured[key] = (out[index] if index is not None else out).detach()
        return hook

    hooks = [module.ln_q_1.register_forward_hook(grab("q1")),
             module.ln_k_1.register_forward_hook(grab("k1")),
             module.ln_v_1.register_forward_hook(grab("v1")),
             module.clip_attn.register_forward_hook(grab("o_tokmajor", 0))]
    with torch.no_grad():
        y = module((x, xm))
    for h in hooks:
        h.remove()
    M = (24 // s) ** 2
    assert y.shape == (B, M, D)
    # clip_attn output is [1, M*B, E] with batch index (region)*B + n  (builder.py:126-134)
    o = captured["o_tokmajor"].reshape(M, B, -1).permute(1, 0, 2).contiguous()

    out = {
        "scale_factor": np.int64(s), "hidden_size": np.int64(D), "batch": np.int64(B),
        "param_seed": np.int64(pseed), "input_seed": np.int64(iseed),
        "out_row_stride": np.int64(ostride), "inter_row_stride": np.int64(INTER_STRIDE),
        "params_sha256": np.array(synth.tensor_digest(*params.values())),
        "inputs_sha256": np.array(synth.tensor_digest(x, xm)),
        "torch_version": np.array(torch.__version__),
        "y": y[:, ::ostride].numpy(),
        "q1": captured["q1"][:, ::INTER_STRIDE].numpy(),
        "k1": captured["k1"][:, ::INTER_STRIDE].numpy(),
        "v1": captured["v1"][:, ::INTER_STRIDE].numpy(),
        "o": o[:, ::INTER_STRIDE].numpy(),
    }

    # the reference's own low-precision behaviour (for "no worse than the reference" gates)
    if D == 256:
        for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            m2 = ref.build_vision_projector(cfg)
            m2.load_state_dict(params)
            m2 = m2.to(dt).eval()
            with torch.no_grad():
                y_lp = m2((x.to(dt), xm.to(dt)))
            out[f"y_ref_{tag}"] = y_lp.float().numpy()

    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: y{tuple(y.shape)} -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


# Adversarial cases (tokenpacker_amd.synth.adversarial_case): name, kind, scale_factor, param seed, input seed.  B = 1,
# D = 256.  Stored: the reference's fp32 output and the reference module's OWN bf16 / fp16 outputs on the same
# (rounded) operands — the yard-stick for "no worse than the reference" gates where fp16/bf16 storage of a
# heavy-tailed or badly conditioned activation dominates the error, whoever computes it.
ADVERSARIAL = [(f"adv_{kind}_s{s}", kind, s, 40 + i, 140 + i)
               for i, (kind, s) in enumerate([("outlier_channels", 2), ("massive_tokens", 2), ("ln_offset", 2),
                                              ("ln_small_var", 2), ("outlier_channels", 3), ("ln_offset", 4)])]


def _ref_module(ref, params, s, D, dtype=torch.float32):
    cfg = type("Cfg", (), {"hidden_size": D, "scale_factor": s})()
    m = ref.build_vision_projector(cfg)
    m.load_state_dict(params, strict=True)
    return m.to(dtype).eval()


def run_adversarial(ref, name, kind, s, pseed, iseed, D=256, B=1):
    params, x, xm = synth.adversarial_case(kind, synth.make_params(pseed, D), *synth.make_inputs(iseed, B))
    out = {"scale_factor": np.int64(s), "hidden_size": np.int64(D), "batch": np.int64(B), "kind": np.array(kind),
           "param_seed": np.int64(pseed), "input_seed": np.int64(iseed),
           "params_sha256": np.array(synth.tensor_digest(*params.values())),
           "inputs_sha256": np.array(synth.tensor_digest(x, xm))}
    with torch.no_grad():
        out["y"] = _ref_module(ref, params, s, D)((x, xm)).numpy()
        for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            p_lp = {k: v.to(dt) for k, v in params.items()}
            y_lp = _ref_module(ref, p_lp, s, D, dt)((x.to(dt), xm.to(dt)))
            # the reference's own low-precision error on the metric the GPU tests use: against exact (fp64) math on
            # the SAME rounded operands
            y_exact = orc.forward(p_lp, x.to(dt), xm.to(dt), scale_factor=s, compute_dtype=torch.float64, io_dtype=dt)
            st = synth.error_stats(y_lp, y_exact)
            out[f"ref_{tag}_err"] = np.array([st["rel_max"], st["rel_l2"], st["p999"]])
            print(f"  {name} reference {tag}: rel_max {st['rel_max']:.3e} rel_l2 {st['rel_l2']:.3e} finite={bool(torch.isfinite(y_lp.float()).all())}")
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


# Backward yard-stick: the reference module's own bf16 / fp16 autograd gradients against fp64 autograd on the same
# rounded operands (per parameter rel-L2), and the pin of the oracle's autograd against the reference's fp32 autograd.
GRAD_CASES = [("grad_s2", 2, 51, 151), ("grad_s3", 3, 52, 152), ("grad_s4", 4, 53, 153)]


def run_grad_yardstick(ref, name, s, pseed, iseed, D=256, B=1):
    params = synth.make_params(pseed, D)
    x, xm = synth.make_inputs(iseed, B)
    M = (24 // s) ** 2
    w = torch.randn(B, M, D, generator=torch.Generator().manual_seed(iseed + 1000))
    names = list(params.keys())

    def ref_grads(p, dt):
        m = _ref_module(ref, p, s, D, dt).train()
        m.zero_grad()
        y = m((x.to(dt), xm.to(dt)))
        (y.float() * w).sum().backward()
        sd = dict(m.named_parameters())
        return [sd[n].grad.detach().double() for n in names]

    def oracle_grads(p):
        pp = {k: v.double().requires_grad_(True) for k, v in p.items()}
        io = next(iter(p.values())).dtype
        y = orc.forward(pp, x.to(io), xm.to(io), scale_factor=s, compute_dtype=torch.float64,
                        io_dtype=None if io == torch.float32 else io)
        (y * w.double()).sum().backward()
        return [pp[n].grad for n in names]

    def rel_all(got, want):
        e = synth.grad_errors(dict(zip(names, got)), dict(zip(names, want)))
        return np.array([e[n] for n in names])

    g_ref32, g_orc = ref_grads(params, torch.float32), oracle_grads(params)
    out = {"scale_factor": np.int64(s), "hidden_size": np.int64(D), "batch": np.int64(B),
           "param_seed": np.int64(pseed), "input_seed": np.int64(iseed), "names": np.array(names),
           "oracle_vs_ref_fp32": rel_all(g_orc, g_ref32)}
    print(f"  {name}: oracle fp64 autograd vs reference fp32 autograd, worst rel-L2 {out['oracle_vs_ref_fp32'].max():.2e}")
    for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        p_lp = {k: v.to(dt) for k, v in params.items()}
        g_lp, g64 = ref_grads(p_lp, dt), oracle_grads(p_lp)
        out[f"ref_{tag}_grad_rel_l2"] = rel_all(g_lp, g64)
        print(f"  {name}: reference {tag} autograd vs fp64, worst rel-L2 {out[f'ref_{tag}_grad_rel_l2'].max():.3e} "
              f"({names[int(out[f'ref_{tag}_grad_rel_l2'].argmax())]})")
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)


def run_mask_case(ref, name="mask_s2_D256_B2", s=2, D=256, B=2, pseed=71, iseed=171):
    """The reference module called WITH an attn_mask (builder.py:107,130): the 2-D float and the 3-D boolean form."""
    params = synth.make_params(pseed, D)
    x, xm = synth.make_inputs(iseed, B)
    m2, m3 = synth.make_attn_masks(iseed + 1, B, s)
    mod = _ref_module(ref, params, s, D)
    with torch.no_grad():
        out = {"scale_factor": np.int64(s), "hidden_size": np.int64(D), "batch": np.int64(B), "param_seed": np.int64(pseed),
               "input_seed": np.int64(iseed), "y_none": mod((x, xm)).numpy(), "y_2d_float": mod((x, xm), attn_mask=m2).numpy(),
               "y_3d_bool": mod((x, xm), attn_mask=m3).numpy(),
               "inputs_sha256": np.array(synth.tensor_digest(x, xm, m2, m3.float()))}
    path = os.path.join(OUT_DIR, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: -> {path} ({os.path.getsize(path) / 1e6:.2f} MB); masks change the output by "
          f"{float(np.abs(out['y_2d_float'] - out['y_none']).max()):.3f} / {float(np.abs(out['y_3d_bool'] - out['y_none']).max()):.3f}")


def main():
    if not os.path.exists(REF_FILE):
        sys.exit(f"{REF_FILE} not found: goldens can only be regenerated in the build container")
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.manual_seed(0)
    ref = load_reference()
    which = sys.argv[1:] or ["cases", "adversarial", "grads", "mask"]
    if "mask" in which:
        run_mask_case(ref)
    if "cases" in which:
        for case in CASES:
            run_case(ref, *case)
    if "adversarial" in which:
        for case in ADVERSARIAL:
            run_adversarial(ref, *case)
    if "grads" in which:
        for case in GRAD_CASES:
            run_grad_yardstick(ref, *case)


if __name__ == "__main__":
    main()
