"""The drop-in inside the reference's own caller, checked where the reference exists (build container, CPU).

``llava/model/llava_arch.py`` is imported UNMODIFIED (``oracle/reference_loader.py``: ``sys.modules`` pre-seeding,
SURVEY.md §8c), with the reference's real ``CLIPVisionTower`` (``clip_encoder.py:7-89``) over a random-init CLIP-L
loaded from a local directory.  The maintainer's 2-line patch of INTEGRATION.md §1 — the name ``TokenPacker`` in
``llava/model/multimodal_projector/builder.py`` bound to ``tokenpacker_amd.TokenPacker`` — is applied by rebinding
that module global; everything else (``LlavaMetaModel.__init__`` :32-34, ``initialize_vision_modules`` :42-83,
``encode_images`` :95-98) runs as written.

There is no GPU here and the product has no CPU path, so VALUES are checked on the GPU box against
``tests/golden/e2e_encode_images.npz`` (``tests/test_gpu_e2e.py``), which this file pins to the unmodified
``encode_images`` by re-minting it.  What this file checks is the wiring: who builds the projector, how its weights
arrive, what tuple the tower hands to ``forward`` (shapes, strides, dtype), and that the call at
``llava_arch.py:97`` lands in the HIP module (it raises the module's own "no CPU fallback" error).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_loader as rl  # noqa: E402
from tokenpacker_amd import TokenPacker, synth  # noqa: E402

pytestmark = pytest.mark.skipif(not rl.reference_available(), reason="needs /root/reference (build container)")

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "e2e_encode_images.npz")


@pytest.fixture(scope="module")
def small_clip_dir(tmp_path_factory):
    """24 layers x 1024 wide like CLIP-L (the projector needs hidden states 12, 16, 22, 23 of width 1024), MLP
    shrunk to 64 so the wiring tests stay fast — hidden-state shapes do not depend on it."""
    return rl.make_clip_dir(str(tmp_path_factory.mktemp("clip_small")), seed=7, intermediate_size=64)


@pytest.fixture()
def patched_builder(monkeypatch):
    rl.import_llava_arch()
    builder = sys.modules["llava.model.multimodal_projector.builder"]
    monkeypatch.setattr(builder, "TokenPacker", TokenPacker)          # the maintainer's patch
    return builder


def test_llava_arch_is_the_unmodified_reference_file():
    arch = rl.import_llava_arch()
    assert os.path.realpath(arch.__file__) == os.path.join(rl.REFERENCE_ROOT, "llava/model/llava_arch.py")
    tower = sys.modules["llava.model.multimodal_encoder.clip_encoder"]
    assert os.path.realpath(tower.__file__).startswith(rl.REFERENCE_ROOT)


def test_hip_module_is_built_loaded_and_called_by_the_reference(small_clip_dir, patched_builder, tmp_path):
    s, D = 3, 256
    params = synth.make_params(71, D)
    adapter = str(tmp_path / "mm_projector.bin")
    torch.save({"model.mm_projector." + k: v for k, v in params.items()}, adapter)    # llava_trainer.py:245-253 naming
    lm, model = rl.build_llava_host(small_clip_dir, D, s, pretrain_mm_mlp_adapter=adapter)

    # llava_arch.py:34 built OUR class through the reference's own factory (builder.py:144-145) ...
    assert type(model.mm_projector) is TokenPacker
    assert (model.mm_projector.scale_factor, model.mm_projector.hidden_size) == (s, D)
    # ... and llava_arch.py:78-83 loaded the checkpoint into it (strict load_state_dict through get_w)
    for k, v in model.mm_projector.state_dict().items():
        assert torch.equal(v, params[k]), k
    # name filters of the training code ("mm_projector" in name: llava_trainer.py:168, train.py:190)
    names = [n for n, _ in model.named_parameters() if "mm_projector" in n]
    assert names == ["mm_projector." + k for k in params]
    # llava_arch.py:75-76 re-enables gradients on an existing projector
    model.mm_projector.requires_grad_(False)
    for p in model.mm_projector.parameters():
        p.requires_grad = True
    assert all(p.requires_grad for p in model.mm_projector.parameters())

    # what the tower hands over at llava_arch.py:96-97: the tuple of NON-contiguous [:, 1:] slices (clip_encoder.py:37-38)
    seen = {}
    hook = model.mm_projector.register_forward_pre_hook(lambda mod, args: seen.setdefault("args", args))
    images = torch.randn(2, 3, 336, 336, generator=torch.Generator().manual_seed(1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # raised by tokenpacker_amd, i.e. the call landed
        lm.encode_images(images)
    hook.remove()
    (feat,) = seen["args"]
    x, xm = feat
    assert isinstance(feat, tuple) and len(feat) == 2
    assert tuple(x.shape) == (2, 576, 1024) and tuple(xm.shape) == (2, 576, 4096)
    assert x.stride() == (577 * 1024, 1024, 1) and xm.stride() == (577 * 4096, 4096, 1)
    assert x.dtype == images.dtype and not xm.is_contiguous()
    # the strided layout is one the C ABI addresses in place (no hidden .contiguous() of 1.2 GB at B = 256)
    assert TokenPacker._addressable(x.to(torch.bfloat16)) is not None
    xb = xm.to(torch.bfloat16)
    assert TokenPacker._addressable(xb) is xb or xb.is_contiguous()
    # select_layer = -2 on 25 hidden states: x is the last quarter of x_multi (SURVEY.md §8a side fact)
    assert torch.equal(x, xm[..., 3072:])


def test_reference_projector_is_what_the_factory_builds_without_the_patch(small_clip_dir):
    rl.import_llava_arch()
    _, model = rl.build_llava_host(small_clip_dir, 256, 2)
    assert type(model.mm_projector).__module__ == "llava.model.multimodal_projector.builder"


def test_e2e_golden_is_what_the_unmodified_encode_images_returns():
    """Re-mint the s = 2 golden (full-size CLIP-L, reference projector, reference encode_images) and compare with
    the committed file: the GPU test's target really is the unmodified reference path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import make_e2e_golden as mk
    z = np.load(GOLDEN)
    assert str(z["torch_version"]) == torch.__version__, "re-mint tests/golden/e2e_encode_images.npz with this torch"
    img = mk.images()
    assert synth.tensor_digest(img) == str(z["images_sha256"])
    import tempfile
    tmp = tempfile.mkdtemp(prefix="tp_clip_")
    clip_dir = rl.make_clip_dir(os.path.join(tmp, "clip"), int(z["clip_seed"]))
    s = 2
    params = synth.make_params(mk.PARAM_SEED[s], int(z["hidden_size"]))
    adapter = os.path.join(tmp, "mm_projector.bin")
    torch.save({"model.mm_projector." + k: v for k, v in params.items()}, adapter)
    lm, model = rl.build_llava_host(clip_dir, int(z["hidden_size"]), s, pretrain_mm_mlp_adapter=adapter)
    with torch.no_grad():
        y = lm.encode_images(img)
        x, xm = model.get_vision_tower()(img)
    assert torch.allclose(y, torch.from_numpy(z["y_s2"]), rtol=0, atol=2e-5 * float(np.abs(z["y_s2"]).max()))
    r, c = int(z["feat_rows"]), int(z["feat_cols"])
    assert torch.allclose(xm[:, ::r, ::c], torch.from_numpy(z["xm_sub"]), atol=1e-4)
    assert torch.allclose(x[:, ::r, ::c], torch.from_numpy(z["x_sub"]), atol=1e-4)
