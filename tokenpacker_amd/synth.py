"""Deterministic synthetic CLIP features and projector parameters.

Used by the tests, ``bench.py``, ``oracle/make_golden.py`` and ``__graft_entry__.smoke()`` so that
every leg (reference module, oracle, HIP path) sees bit-identical inputs.  Everything is generated
on the CPU from an explicit ``torch.Generator`` — nothing here touches the GPU or the oracle.

Shapes follow SURVEY.md §8(d): ``x`` is the select-layer CLIP grid ``[B, 576, 1024]`` and
``x_multi`` the four concatenated layers ``[B, 576, 4096]`` that
``CLIPVisionTower.forward`` returns (reference ``llava/model/multimodal_encoder/clip_encoder.py:28-44,62``).
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import torch

RAW_GRID = 24          # CLIP-L/14 @ 336 px
N_TOKENS = RAW_GRID * RAW_GRID
C_CLIP = 1024          # CLIP width == embed_dim == kv_dim
C_MULTI = 4096         # 4 CLIP layers concatenated
N_HEADS = 8


def param_shapes(hidden_size: int) -> "OrderedDict[str, tuple]":
    """State-dict names/shapes of the projector, in the reference's registration order
    (reference ``llava/model/multimodal_projector/builder.py:40-85``)."""
    D = hidden_size
    E = C_CLIP
    return OrderedDict([
        ("q_proj_1.weight", (E, E)),
        ("k_proj_1.0.weight", (E, C_MULTI)), ("k_proj_1.0.bias", (E,)),
        ("k_proj_1.2.weight", (E, E)), ("k_proj_1.2.bias", (E,)),
        ("v_proj_1.0.weight", (E, C_MULTI)), ("v_proj_1.0.bias", (E,)),
        ("v_proj_1.2.weight", (E, E)), ("v_proj_1.2.bias", (E,)),
        ("ln_q_1.weight", (E,)), ("ln_q_1.bias", (E,)),
        ("ln_k_1.weight", (E,)), ("ln_k_1.bias", (E,)),
        ("ln_v_1.weight", (E,)), ("ln_v_1.bias", (E,)),
        ("clip_attn.in_proj_weight", (3 * E, E)), ("clip_attn.in_proj_bias", (3 * E,)),
        ("clip_attn.out_proj.weight", (E, E)), ("clip_attn.out_proj.bias", (E,)),
        ("mlp.0.weight", (D, E)), ("mlp.0.bias", (D,)),
        ("mlp.2.weight", (D, D)), ("mlp.2.bias", (D,)),
    ])


def make_params(seed: int, hidden_size: int = 4096, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Non-trivial parameters: the reference's default init has zero biases and unit LayerNorm
    affine, which would hide bias / affine bugs.

    Linear weights ~ N(0, 1/fan_in) * 1.5 so activations stay O(1) through the chain (the
    reference's trunc_normal(std=.02) gives tiny pre-LN activations; LN rescales them anyway),
    biases ~ N(0, 0.1), LN gamma ~ 1 + N(0, 0.1), LN beta ~ N(0, 0.1).
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = OrderedDict()
    for name, shape in param_shapes(hidden_size).items():
        if name.startswith("ln_") and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1]
            t = torch.randn(shape, generator=g) * (1.5 / fan_in ** 0.5)
        out[name] = t.to(dtype)
    return out


def make_inputs(seed: int, B: int, dtype=torch.float32, layout: str = "contiguous",
                tie_select_layer: bool = False):
    """Synthetic ``(x, x_multi)``.

    layout="contiguous": plain ``[B,576,C]`` tensors (eval path: fp16 tower -> bf16 cast copies).
    layout="tower": both are ``[:, 1:]`` slices of CLS-prefixed ``[B,577,C]`` buffers, i.e. the
    non-contiguous view the tower hands over when tower dtype == image dtype
    (reference ``clip_encoder.py:37-38,62``; SURVEY.md §7 "hard parts").
    tie_select_layer: x == x_multi[..., 3072:] as with ``mm_vision_select_layer=-2``.
    The VALUES do not depend on ``layout``.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    xm = torch.randn(B, N_TOKENS, C_MULTI, generator=g)
    x = torch.randn(B, N_TOKENS, C_CLIP, generator=g)
    if tie_select_layer:
        x = xm[..., C_MULTI - C_CLIP:].clone()
    x = x.to(dtype)
    xm = xm.to(dtype)
    if layout == "tower":
        xb = torch.zeros(B, N_TOKENS + 1, C_CLIP, dtype=dtype)
        xmb = torch.zeros(B, N_TOKENS + 1, C_MULTI, dtype=dtype)
        xb[:, 1:] = x
        xmb[:, 1:] = xm
        x, xm = xb[:, 1:], xmb[:, 1:]
    elif layout != "contiguous":
        raise ValueError(f"unknown layout {layout!r}")
    return x, xm


def tensor_digest(*tensors: torch.Tensor) -> str:
    """sha256 over the raw bytes (after .contiguous()); pins RNG streams in the golden files."""
    h = hashlib.sha256()
    for t in tensors:
        t = t.detach().contiguous().cpu()
        h.update(str(t.dtype).encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------------
# Adversarial variants (heavy-tailed features, LayerNorm corner cases).  Trained CLIP-L hidden states are not
# unit-normal: a handful of channels carry activations in the hundreds ("massive activations"), concentrated in
# a few tokens.  The HIP path keeps every inter-kernel activation in fp16, so these regimes are
# tested explicitly — against the fp64 oracle AND against the reference module's own low-precision error
# (goldens minted by oracle/make_golden.py carry the reference's bf16 / fp16 outputs on the same cases).
ADVERSARIAL_KINDS = ("outlier_channels", "massive_tokens", "ln_offset", "ln_small_var")


def adversarial_case(kind: str, params: "OrderedDict[str, torch.Tensor]", x: torch.Tensor, xm: torch.Tensor):
    """Returns modified copies ``(params, x, x_multi)`` (fp32 in, fp32 out; cast afterwards).

    outlier_channels : 6 channels of x_multi / 2 of x scaled x300 in every token.
    massive_tokens   : 5 tokens per image carry +-400 in 3 channels of x_multi and x (sign alternating).
    ln_offset        : k/v second-layer biases shifted by +40 — LayerNorm rows with |mean| >> std
                       (one-pass variance and the folded mean term cancel catastrophically there).
    ln_small_var     : k/v second layers scaled down to a ~1e-2 spread around a constant bias — rstd ~ 50-100
                       amplifies every rounding error made before the LayerNorm."""
    p = OrderedDict((k, v.clone()) for k, v in params.items())
    x, xm = x.clone(), xm.clone()
    if kind == "outlier_channels":
        xm[..., [7, 1029, 2051, 3073, 3500, 4090]] *= 300.0
        x[..., [1, 1000]] *= 300.0
    elif kind == "massive_tokens":
        tok = torch.tensor([0, 23, 300, 301, 575])
        sgn = torch.tensor([1.0, -1.0, 1.0, -1.0, 1.0]).view(1, 5, 1)
        for ch in (5, 2500, 4000):
            xm[:, tok, ch] = (400.0 * sgn).expand(xm.shape[0], 5, 1)[..., 0]
        for ch in (5, 900):
            x[:, tok, ch] = (400.0 * sgn).expand(x.shape[0], 5, 1)[..., 0]
    elif kind == "ln_offset":
        p["k_proj_1.2.bias"] += 40.0
        p["v_proj_1.2.bias"] -= 40.0
    elif kind == "ln_small_var":
        for br in ("k_proj_1", "v_proj_1"):
            p[br + ".2.weight"] *= 1e-2
            p[br + ".2.bias"] = 1.0 + 1e-2 * p[br + ".2.bias"]
    else:
        raise ValueError(f"unknown adversarial kind {kind!r}")
    return p, x, xm


def error_stats(y: torch.Tensor, y_ref: torch.Tensor) -> dict:
    """Parity metrics reported by the GPU tests: the global-max metric of SURVEY.md §8c, rel-L2, and the
    99.9th percentile of the element-wise error normalised by max|y_ref| (hides nothing behind one large
    reference element) and by the element's own magnitude floor-ed at 1 % of max|y_ref|."""
    y = y.detach().to(torch.float64).cpu().reshape(-1)
    r = y_ref.detach().to(torch.float64).cpu().reshape(-1)
    d = (y - r).abs()
    scale = float(r.abs().max()) + 1e-300
    k = max(int(0.999 * d.numel()), 1)
    p999 = float(d.kthvalue(min(k, d.numel())).values)
    rel_elem = d / r.abs().clamp_min(1e-2 * scale)
    return {"rel_max": float(d.max()) / scale, "rel_l2": float(d.norm() / (r.norm() + 1e-300)),
            "p999": p999 / scale, "p999_elem": float(rel_elem.kthvalue(min(k, d.numel())).values)}


def grad_errors(got: dict, want: dict) -> dict:
    """Per-parameter gradient error used by the backward tests and by the reference yard-stick minted in
    oracle/make_golden.py: rms(g - g_ref) / max(rms(g_ref), 0.1 * largest rms among same-shaped parameters).
    Some gradients are mathematically ZERO (ln_k_1.bias and the k-third of in_proj_bias shift every logit of a
    region by the same amount, which softmax ignores); their computed value is the round-off of cancelling terms as
    large as the other gradients of the same shape, hence the floor."""
    rms = {k: float(v.double().norm()) / v.numel() ** 0.5 for k, v in want.items()}
    out = {}
    for k, w in want.items():
        scale = max(rms[k], 0.1 * max(rms[j] for j in want if want[j].shape == w.shape))
        out[k] = float((got[k].double() - w.double()).norm()) / w.numel() ** 0.5 / (scale + 1e-300)
    return out


def make_attn_masks(seed: int, B: int, s: int, raw_grid: int = RAW_GRID):
    """Two attn_mask arguments for ``TokenPacker.forward(x, attn_mask)`` in the forms nn.MultiheadAttention accepts:
    a 2-D additive float mask ``[1, s*s]`` and a 3-D boolean mask ``[(M*B)*8, 1, s*s]`` (True = masked out; at least
    one key of every row stays visible)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    S2, M = s * s, (raw_grid // s) ** 2
    m2 = 2.0 * torch.randn(1, S2, generator=g)
    m3 = torch.rand(M * B * N_HEADS, 1, S2, generator=g) < 0.35
    m3[:, 0, 0] = False
    return m2, m3
