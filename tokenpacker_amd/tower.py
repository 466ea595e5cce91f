"""Producer-side glue: the feature selection of the reference's CLIP tower WITHOUT its
``torch.cat``.

``CLIPVisionTower.feature_select`` (reference ``llava/model/multimodal_encoder/clip_encoder.py:28-44``) picks the
hidden state ``select_layer`` as ``x`` and concatenates hidden states 12, 16, 22, 23 along the channel axis into
``x_multi`` (1.2 GB written and read again at B = 256), then drops the CLS token with ``[:, 1:]``.
:func:`select_features` returns the same ``x`` and the FOUR ``[:, 1:]`` views instead; ``TokenPacker.forward``
accepts ``(x, (p0, p1, p2, p3))`` and walks the four sources as K-ranges of its first GEMM
(``tp_forward_parts``), bit-identical to the concatenated form.  With the shipped ``--mm_vision_select_layer -2``
``x`` is the same tensor as the last part (hidden state 23 of 25), so nothing is copied at all.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

MULTI_LAYERS = (12, 16, 22, 23)        # clip_encoder.py:28


def select_features(hidden_states: Sequence[torch.Tensor], select_layer: int = -2,
                    layers: Sequence[int] = MULTI_LAYERS, select_feature: str = "patch"
                    ) -> Tuple[torch.Tensor, Tuple[torch.Tensor, ...]]:
    """``hidden_states``: the tower's per-layer outputs ``[B, 577, 1024]`` (``output_hidden_states=True``).
    Returns ``(x, parts)`` — views, no copies — to be passed to the projector as ``(x, parts)``."""
    if select_feature == "patch":
        cut = slice(1, None)           # drop CLS (clip_encoder.py:37-38)
    elif select_feature == "cls_patch":
        cut = slice(None)
    else:
        raise ValueError(f"Unexpected select feature: {select_feature}")      # clip_encoder.py:43
    parts = tuple(hidden_states[l][:, cut] for l in layers)
    return hidden_states[select_layer][:, cut], parts


def concat_reference(hidden_states: Sequence[torch.Tensor], select_layer: int = -2,
                     layers: Sequence[int] = MULTI_LAYERS) -> Tuple[torch.Tensor, torch.Tensor]:
    """What the reference's ``feature_select`` returns (with the ``torch.cat``) — for tests and comparisons."""
    multi = torch.cat([hidden_states[l] for l in layers], dim=2)
    return hidden_states[select_layer][:, 1:], multi[:, 1:]
