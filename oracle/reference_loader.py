"""Loaders for the UNMODIFIED reference sources.  TEST INFRASTRUCTURE ONLY (build container).

``/root/reference`` exists in the build container only (never on the GPU box), so everything here
is used by the golden-minting scripts (``oracle/make_*golden.py``) and by the ``-m "not gpu"``
tests; nothing under ``tokenpacker_amd/`` imports it.  No reference source is copied: the files
are imported where they lie.

* :func:`load_builder` — ``llava/model/multimodal_projector/builder.py`` by file path.
* :func:`import_llava_arch` — the caller of the hot path, ``llava/model/llava_arch.py`` (its
  ``encode_images`` at :95-98 is the drop-in boundary).  ``import llava`` itself fails under the
  installed transformers (``llava/__init__.py:1`` -> ``llava_llama.py:142`` re-registers the
  ``llava`` config name), so ``sys.modules['llava']`` / ``['llava.model']`` are pre-seeded with
  empty packages whose ``__path__`` points at the reference directories; the
  sub-modules ``llava.constants``, ``llava.model.multimodal_encoder.*``,
  ``llava.model.multimodal_projector.builder`` and ``llava.model.llava_arch`` are then imported
  from the reference files, unmodified.
* :func:`make_clip_dir` — a random-init CLIP-ViT-L/14-336 (``transformers.CLIPVisionModel``)
  saved as a local "pretrained" directory, so that the reference's own ``CLIPVisionTower``
  (``clip_encoder.py:7-89``) loads it through ``from_pretrained`` without a network.
* :func:`build_llava_host` — the smallest model the reference's mixins accept:
  ``class Host(LlavaMetaModel, nn.Module)`` + ``class HostLM(LlavaMetaForCausalLM)``.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types
from typing import Optional

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"
BUILDER_FILE = os.path.join(REFERENCE_ROOT, "llava/model/multimodal_projector/builder.py")


def reference_available() -> bool:
    return os.path.exists(BUILDER_FILE)


def load_builder():
    """The reference projector module, imported by file path (nothing else of llava is touched)."""
    spec = importlib.util.spec_from_file_location("_ref_builder", BUILDER_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_llava_arch():
    """``llava.model.llava_arch`` of the reference, unmodified (see module docstring)."""
    if "llava.model.llava_arch" in sys.modules:
        return sys.modules["llava.model.llava_arch"]
    for name, rel in (("llava", "llava"), ("llava.model", "llava/model")):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
            pkg.__package__ = name
            sys.modules[name] = pkg
    sys.modules["llava"].model = sys.modules["llava.model"]
    return importlib.import_module("llava.model.llava_arch")


def clip_l_config(intermediate_size: int = 4096):
    """CLIP-ViT-L/14 @ 336 px vision config (the tower every reference script names,
    ``scripts/v1_5/pretrain.sh:11``): 24 layers x 1024 wide, 577 tokens.  ``intermediate_size`` can be
    shrunk for wiring tests — the projector only sees hidden states, whose shape does not depend on it."""
    from transformers import CLIPVisionConfig
    return CLIPVisionConfig(hidden_size=1024, intermediate_size=intermediate_size, num_hidden_layers=24,
                            num_attention_heads=16, image_size=336, patch_size=14)


def make_clip(seed: int, intermediate_size: int = 4096):
    """Random-init CLIP vision model, deterministic in (seed, torch version, transformers version)."""
    from transformers import CLIPVisionModel
    torch.manual_seed(seed)
    return CLIPVisionModel(clip_l_config(intermediate_size)).eval().requires_grad_(False)


def make_clip_dir(path: str, seed: int, intermediate_size: int = 4096) -> str:
    """Save :func:`make_clip` + an image-processor config under ``path`` (a local 'pretrained' dir)."""
    from transformers import CLIPImageProcessor
    os.makedirs(path, exist_ok=True)
    make_clip(seed, intermediate_size).save_pretrained(path)
    CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}).save_pretrained(path)
    return path


class _Base(nn.Module):
    """What ``LlavaMetaModel.__init__`` needs from its sibling base class (``LlamaModel`` in the
    reference, ``llava_llama.py:35-39``): a constructor taking the config, and ``embed_tokens``."""

    def __init__(self, config):
        nn.Module.__init__(self)
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)


def build_llava_host(clip_dir: str, hidden_size: int, scale_factor: int, vocab_size: int = 64,
                     pretrain_mm_mlp_adapter: Optional[str] = None, select_layer: int = -2):
    """Instantiate the reference's ``LlavaMetaModel`` / ``LlavaMetaForCausalLM`` mixins around a stub LM and
    run ``initialize_vision_modules`` (``llava_arch.py:42-83``) exactly as ``train.py:935`` does.
    Returns ``(lm, model)``; ``lm.encode_images(images)`` is the unmodified ``llava_arch.py:95-98``."""
    arch = import_llava_arch()

    class Host(arch.LlavaMetaModel, _Base):
        pass

    class HostLM(arch.LlavaMetaForCausalLM, nn.Module):
        def __init__(self, model):
            nn.Module.__init__(self)
            self.model = model
            self.config = model.config

        def get_model(self):
            return self.model

        @property
        def device(self):
            return next(self.model.parameters()).device

    cfg = types.SimpleNamespace(hidden_size=hidden_size, scale_factor=scale_factor, vocab_size=vocab_size,
                                mm_vision_tower=clip_dir, mm_vision_select_layer=select_layer,
                                mm_vision_select_feature="patch", mm_projector_type="tokenpacker")
    model = Host(cfg)                       # llava_arch.py:32-34: tower (delay_load) + build_vision_projector(config)
    args = types.SimpleNamespace(vision_tower=clip_dir, mm_vision_select_layer=select_layer,
                                 mm_vision_select_feature="patch", pretrain_mm_mlp_adapter=pretrain_mm_mlp_adapter,
                                 mm_projector_type="tokenpacker", scale_factor=scale_factor)
    model.initialize_vision_modules(args)   # llava_arch.py:42-83: load_model(), config fields, load_state_dict(get_w(...))
    return HostLM(model), model
