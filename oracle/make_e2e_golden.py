#!/usr/bin/env python3
"""Mint ``tests/golden/e2e_encode_images.npz``: the reference's UNMODIFIED ``encode_images()``
(``llava/model/llava_arch.py:95-98``) run end to end on CPU — the reference's own ``CLIPVisionTower``
(``clip_encoder.py:7-89``, a random-init CLIP-ViT-L/14-336 loaded through ``from_pretrained`` from a local
directory) feeding the reference ``TokenPacker`` whose weights arrive through the reference's own
``initialize_vision_modules`` / ``load_state_dict(get_w(...))`` path (``llava_arch.py:42-83``) from an
``mm_projector.bin`` written the way ``llava_trainer.py:239-256`` writes it.

Build container only (``/root/reference`` is absent on the GPU box).  What travels is this file's output: the GPU
test rebuilds the SAME tower from the seed with plain ``transformers`` (no reference code), runs the same images
through it, hands the ``[:, 1:]`` feature slices to the HIP projector and compares with the ``y`` stored here.
Sub-sampled tower features are stored too, so a drift of the tower (RNG / library version) is told apart from a
projector error.

    python oracle/make_e2e_golden.py
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_loader as rl  # noqa: E402
from tokenpacker_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "e2e_encode_images.npz")
CLIP_SEED, IMAGE_SEED, B, D = 2024, 77, 2, 256
PARAM_SEED = {2: 61, 3: 62, 4: 63}
FEAT_ROWS, FEAT_COLS = 48, 64            # feature sub-sampling strides


def images(seed: int = IMAGE_SEED, batch: int = B) -> torch.Tensor:
    return torch.randn(batch, 3, 336, 336, generator=torch.Generator().manual_seed(seed))


def main():
    if not rl.reference_available():
        sys.exit("reference not found: this golden can only be minted in the build container")
    tmp = tempfile.mkdtemp(prefix="tp_clip_")
    clip_dir = rl.make_clip_dir(os.path.join(tmp, "clip"), CLIP_SEED)
    img = images()
    out = {"clip_seed": np.int64(CLIP_SEED), "image_seed": np.int64(IMAGE_SEED), "batch": np.int64(B),
           "hidden_size": np.int64(D), "torch_version": np.array(torch.__version__),
           "feat_rows": np.int64(FEAT_ROWS), "feat_cols": np.int64(FEAT_COLS),
           "images_sha256": np.array(synth.tensor_digest(img))}
    for s, pseed in PARAM_SEED.items():
        params = synth.make_params(pseed, D)
        adapter = os.path.join(tmp, f"mm_projector_s{s}.bin")
        torch.save({"model.mm_projector." + k: v for k, v in params.items()}, adapter)    # llava_trainer.py:245-253 naming
        lm, model = rl.build_llava_host(clip_dir, D, s, pretrain_mm_mlp_adapter=adapter)
        assert type(model.mm_projector).__module__.endswith("multimodal_projector.builder")
        for k, v in model.mm_projector.state_dict().items():
            assert torch.equal(v, params[k]), k
        with torch.no_grad():
            y = lm.encode_images(img)                                  # llava_arch.py:95-98, unmodified
            x, xm = model.get_vision_tower()(img)
        assert y.shape == (B, (24 // s) ** 2, D) and not xm.is_contiguous()
        out[f"y_s{s}"] = y.numpy()
        if s == 2:
            out["x_sub"] = x[:, ::FEAT_ROWS, ::FEAT_COLS].contiguous().numpy()
            out["xm_sub"] = xm[:, ::FEAT_ROWS, ::FEAT_COLS].contiguous().numpy()
            out["feat_std"] = np.float64(float(xm.std()))
        # the reference in low precision on the same features (what eval does: model .to(bf16), model_vqa_loader.py:134)
        for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            proj = model.mm_projector.to(dt)
            with torch.no_grad():
                y_lp = proj((x.to(dt), xm.to(dt)))
            out[f"ref_{tag}_err_s{s}"] = np.float64(synth.error_stats(y_lp, y)["rel_max"])
            model.mm_projector.to(torch.float32)
            model.mm_projector.load_state_dict(params)
        print(f"s={s}: y{tuple(y.shape)}  reference bf16 / fp16 rel_max vs fp32: "
              f"{float(out[f'ref_bf16_err_s{s}']):.3e} / {float(out[f'ref_fp16_err_s{s}']):.3e}")
    np.savez_compressed(OUT, **out)
    print(f"-> {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
