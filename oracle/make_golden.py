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

import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenpacker_amd import synth  # noqa: E402

REF_FILE = "/root/reference/llava/model/multimodal_projector/builder.py"
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
    spec = importlib.util.spec_from_file_location("_ref_builder", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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


def main():
    if not os.path.exists(REF_FILE):
        sys.exit(f"{REF_FILE} not found: goldens can only be regenerated in the build container")
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.manual_seed(0)
    ref = load_reference()
    for case in CASES:
        run_case(ref, *case)


if __name__ == "__main__":
    main()
