"""tokenpacker_amd — MI355X-native (gfx950) TokenPacker region-to-point visual projector.

Public surface mirrors the reference's ``llava/model/multimodal_projector/builder.py``:
``TokenPacker`` and ``build_vision_projector``; plus the batch-shard helper for one-process-per-GPU
runs (``tokenpacker_amd.shard``).  Importing this package needs neither a GPU nor the built
library; *using* the projector needs both.
"""
from .projector import TokenPacker, build_vision_projector  # noqa: F401

__all__ = ["TokenPacker", "build_vision_projector"]
__version__ = "0.1.0"
