"""CPU oracle for the TokenPacker region-to-point projector.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (index/einsum form, torch CPU tensors as plain arrays)
of the reference's algorithm for the hot path
``TokenPacker.forward`` — reference ``llava/model/multimodal_projector/builder.py:107-137``.
It is NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the timed CPU
baseline.  The product path (``tokenpacker_amd``) never imports it and fails loudly when the HIP
library is missing.

Parity status: **pinned**.  The reference ships no tests or golden vectors for this path
(SURVEY.md §4, §8c), so the pin is the reference module itself, imported by file path in the
build container and run on CPU in fp32 by ``oracle/make_golden.py``; its outputs and sub-sampled
intermediates are committed under ``tests/golden/`` and ``tests/test_oracle_golden.py`` checks
this restatement against every one of them.  The arithmetic the reference delegates to PyTorch
(un-pinned third-party dependency, reference ``pyproject.toml:15-21``; torch 2.10.0 here) is
restated below from its published semantics:

* ``F.interpolate(mode='bilinear', align_corners=False)``  -> :func:`bilinear_downsample`
* ``nn.LayerNorm(eps=1e-6)`` (biased variance)             -> :func:`layer_norm`
* ``nn.GELU()`` (exact erf form)                            -> :func:`gelu_erf`
* ``nn.MultiheadAttention`` with L=1 query, S=s*s keys, need_weights=True explicit path
  (``torch/nn/functional.py:6206-6618``)                    -> :func:`region_attention`
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

EPS = 1e-6          # builder.py:48  norm_layer=partial(nn.LayerNorm, eps=1e-6)
HEADS = 8           # builder.py:44  num_heads=1024//128


# --------------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------------
def gelu_erf(t: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form (builder.py:63,69,81)."""
    return 0.5 * t * (1.0 + torch.erf(t * (1.0 / math.sqrt(2.0))))


def layer_norm(t: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """Per-row LayerNorm over the last axis, biased variance (builder.py:73-75)."""
    mu = t.mean(dim=-1, keepdim=True)
    var = ((t - mu) ** 2).mean(dim=-1, keepdim=True)
    return (t - mu) / torch.sqrt(var + eps) * gamma + beta


def linear(t: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = t @ w^T (+ b): nn.Linear weight layout [out, in]."""
    y = t @ w.transpose(-1, -2)
    return y if b is None else y + b


def bilinear_downsample(grid: torch.Tensor, out_hw: int) -> torch.Tensor:
    """Generic bilinear resize of ``grid[B, H, W, C]`` to ``[B, out, out, C]`` with
    align_corners=False, antialias=False (the op at builder.py:117).

    Source coordinate of output pixel i:  src = (i + 0.5) * (H / out) - 0.5, clamped at 0;
    i0 = floor(src), i1 = min(i0 + 1, H - 1), weight of i1 = src - i0.
    """
    B, H, W, C = grid.shape

    def taps(n_in: int, n_out: int):
        scale = n_in / n_out
        src = (torch.arange(n_out, dtype=torch.float64) + 0.5) * scale - 0.5
        src = src.clamp(min=0.0)
        i0 = src.floor().to(torch.int64).clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        w1 = (src - i0.to(torch.float64)).to(grid.dtype)
        return i0, i1, w1

    r0, r1, wr = taps(H, out_hw)
    c0, c1, wc = taps(W, out_hw)
    top = grid[:, r0] * (1 - wr)[None, :, None, None] + grid[:, r1] * wr[None, :, None, None]
    out = top[:, :, c0] * (1 - wc)[None, None, :, None] + top[:, :, c1] * wc[None, None, :, None]
    return out


def point_queries_closed_form(grid: torch.Tensor, s: int) -> torch.Tensor:
    """The exact integer-ratio degenerations of the bilinear downsample:
    s=2 -> mean of each 2x2 block; s=3 -> centre pixel (3i+1, 3j+1);
    s=4 -> mean of the inner 2x2 (rows 4i+1..4i+2, cols 4j+1..4j+2).  grid: [B, g, g, C]."""
    B, g, _, C = grid.shape
    G = g // s
    blk = grid.reshape(B, G, s, G, s, C)
    if s == 1:
        return grid
    if s == 2:
        return blk.mean(dim=(2, 4))
    if s == 3:
        return blk[:, :, 1, :, 1]
    if s == 4:
        return blk[:, :, 1:3, :, 1:3].mean(dim=(2, 4))
    raise ValueError("closed form only stated for s in {1,2,3,4}")


def point_queries(x: torch.Tensor, raw_grid: int, s: int, io_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Coarse point queries q0[B, M, C] (builder.py:117-118).  The reference always does this
    step in fp32 and casts back to the input dtype; ``io_dtype`` (e.g. torch.bfloat16)
    reproduces that final rounding when the oracle simulates a low-precision module."""
    B, N, C = x.shape
    G = raw_grid // s
    q = bilinear_downsample(x.reshape(B, raw_grid, raw_grid, C), G).reshape(B, G * G, C)
    if io_dtype is not None:
        q = q.to(io_dtype).to(x.dtype)
    return q


def region_gather(t: torch.Tensor, raw_grid: int, s: int) -> torch.Tensor:
    """Region partition = the reference's ``divide_feature`` (builder.py:96-105) in index form.
    t: [B, g*g, c] -> [B, G, G, s*s, c]; region (i, j) holds tokens (i*s+a, j*s+b), key index
    a*s+b."""
    B, N, c = t.shape
    G = raw_grid // s
    return (t.reshape(B, G, s, G, s, c).permute(0, 1, 3, 2, 4, 5).reshape(B, G, G, s * s, c))


def mask_to_regions(attn_mask: torch.Tensor, B: int, G: int, s: int, heads: int = HEADS) -> torch.Tensor:
    """``attn_mask`` as nn.MultiheadAttention takes it from builder.py:126-130 — 2-D ``[1, s*s]`` or 3-D
    ``[(M*B)*heads, 1, s*s]`` with the batch index ``region * B + image`` that ``divide_feature`` produces
    (builder.py:96-105); boolean (True = masked out) or additive float — as an additive ``[B, G, G, heads, s*s]``."""
    m = attn_mask
    if m.dtype == torch.bool:
        m = torch.zeros(m.shape, dtype=torch.float64).masked_fill(m, float("-inf"))
    m = m.to(torch.float64)
    S2, M = s * s, G * G
    if m.dim() == 2:
        return m.reshape(1, 1, 1, 1, S2).expand(B, G, G, heads, S2)
    return m.reshape(M, B, heads, S2).permute(1, 0, 2, 3).reshape(B, G, G, heads, S2)


def region_attention(q1, k1, v1, in_w, in_b, raw_grid: int, s: int, heads: int = HEADS, attn_mask=None):
    """Region-to-point cross attention (builder.py:122-130): every coarse query attends to the
    s*s fine tokens of its own region, 8 heads of d=128, softmax over the s*s keys, scale
    1/sqrt(d) applied to q.  Returns the concatenated heads [B, M, E] BEFORE out_proj."""
    B, M, E = q1.shape
    d = E // heads
    G = raw_grid // s
    wq, wk, wv = in_w[:E], in_w[E:2 * E], in_w[2 * E:]
    bq, bk, bv = in_b[:E], in_b[E:2 * E], in_b[2 * E:]
    Q = linear(q1, wq, bq).reshape(B, G, G, heads, d) * (1.0 / math.sqrt(d))
    K = region_gather(linear(k1, wk, bk), raw_grid, s).reshape(B, G, G, s * s, heads, d)
    V = region_gather(linear(v1, wv, bv), raw_grid, s).reshape(B, G, G, s * s, heads, d)
    logits = torch.einsum("bijhd,bijkhd->bijhk", Q, K)
    if attn_mask is not None:
        logits = logits + mask_to_regions(attn_mask, B, G, s, heads).to(logits.dtype)
    P = torch.softmax(logits, dim=-1)
    O = torch.einsum("bijhk,bijkhd->bijhd", P, V)
    return O.reshape(B, M, E)


# --------------------------------------------------------------------------------------------
# the whole path
# --------------------------------------------------------------------------------------------
def forward(params: Dict[str, torch.Tensor], x: torch.Tensor, x_multi: torch.Tensor,
            scale_factor: int = 2, raw_grid: int = 24,
            compute_dtype: torch.dtype = torch.float64,
            io_dtype: Optional[torch.dtype] = None,
            return_intermediates: bool = False, attn_mask=None):
    """Oracle for ``TokenPacker.forward((x, x_multi))`` (builder.py:107-137).

    ``params`` uses the reference's state-dict names.  Inputs/weights are up-cast to
    ``compute_dtype`` (fp64 = clean reference for error measurement, fp32 = the reference's CPU
    arithmetic / the timed CPU baseline).  The result is what exact arithmetic on the *given
    (possibly bf16/fp16-rounded)* weights and inputs yields — the ``y32`` of SURVEY.md §8(c).
    """
    if raw_grid % scale_factor != 0:
        raise ValueError("scale_factor must be divisible by grid size")   # builder.py:51-52
    p = {k: v.to(compute_dtype) for k, v in params.items()}
    x = x.to(compute_dtype)
    xm = x_multi.to(compute_dtype)
    s = scale_factor

    # K / V branches over the multi-level features (builder.py:112-113)
    k1 = layer_norm(linear(gelu_erf(linear(xm, p["k_proj_1.0.weight"], p["k_proj_1.0.bias"])),
                           p["k_proj_1.2.weight"], p["k_proj_1.2.bias"]),
                    p["ln_k_1.weight"], p["ln_k_1.bias"])
    v1 = layer_norm(linear(gelu_erf(linear(xm, p["v_proj_1.0.weight"], p["v_proj_1.0.bias"])),
                           p["v_proj_1.2.weight"], p["v_proj_1.2.bias"]),
                    p["ln_v_1.weight"], p["ln_v_1.bias"])
    # coarse point queries (builder.py:117-120)
    q0 = point_queries(x, raw_grid, s, io_dtype)
    q1 = layer_norm(linear(q0, p["q_proj_1.weight"]), p["ln_q_1.weight"], p["ln_q_1.bias"])
    # region-to-point attention (builder.py:122-130) and out_proj
    attn = region_attention(q1, k1, v1, p["clip_attn.in_proj_weight"], p["clip_attn.in_proj_bias"],
                            raw_grid, s, attn_mask=attn_mask)
    o = linear(attn, p["clip_attn.out_proj.weight"], p["clip_attn.out_proj.bias"])
    # output MLP (builder.py:136)
    y = linear(gelu_erf(linear(o, p["mlp.0.weight"], p["mlp.0.bias"])),
               p["mlp.2.weight"], p["mlp.2.bias"])
    if return_intermediates:
        return y, {"q0": q0, "q1": q1, "k1": k1, "v1": v1, "attn": attn, "o": o}
    return y


def rel_err(y: torch.Tensor, y_ref: torch.Tensor) -> float:
    """Parity metric of SURVEY.md §8(c): max|y - y_ref| / max|y_ref|."""
    y = y.detach().to(torch.float64).cpu()
    y_ref = y_ref.detach().to(torch.float64).cpu()
    return float((y - y_ref).abs().max() / y_ref.abs().max())


def rel_l2(y: torch.Tensor, y_ref: torch.Tensor) -> float:
    y = y.detach().to(torch.float64).cpu()
    y_ref = y_ref.detach().to(torch.float64).cpu()
    return float((y - y_ref).norm() / y_ref.norm())
