"""Helpers for the -m gpu parity tests: thin wrappers over the C ABI and failure diagnostics."""
from __future__ import annotations

import ctypes

import torch

from tokenpacker_amd import _capi

DT = {torch.bfloat16: _capi.TP_BF16, torch.float16: _capi.TP_F16, torch.float32: _capi.TP_F32}


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def linear(A, W, bias=None, flags=0, tile=0, rows_per_batch=0, a_batch_stride=0, lda=None, M=None,
           mean_rstd=None, colsum=None, want_stats=False, out_dtype=None, half=False, t192=False, sync=True):
    """C = epilogue(A · W^T) through tp_linear.  A may be a 2-D tensor or a raw (ptr-bearing) tensor
    with explicit M / lda / batch strides.  tile: 0 auto | 128 (the 128-tile kernel, tp_gemm.hip) | 256 (the ping-pong kernel,
    tp_gemm8.hip); half=True: every tile of the ping-pong kernel a 128 x 256 half tile (TP_TUNE_GEMM_TILE = 2); t192=True: 192 x 256
    tiles (TP_TUNE_GEMM_TILE = 3; plain launches)."""
    lib = _capi.load_library()
    N, K = W.shape
    if M is None:
        M = A.shape[0]
    if lda is None:
        lda = A.stride(0)
    out_dtype = W.dtype if out_dtype is None else out_dtype
    C = torch.empty(M, N, dtype=out_dtype, device=W.device)
    args = _capi.tp_linear_args()
    args.M, args.N, args.K = M, N, K
    args.dtype = DT[W.dtype]
    args.out_dtype = DT[out_dtype]
    args.flags = flags | (_capi.TP_LINEAR_ROW_STATS if want_stats else 0)
    args.rows_per_batch = rows_per_batch
    args.a_batch_stride = a_batch_stride
    args.lda, args.ldc = lda, N
    args.A, args.W, args.C = A.data_ptr(), W.data_ptr(), C.data_ptr()
    args.bias = bias.data_ptr() if bias is not None else None
    args.row_mean_rstd = mean_rstd.data_ptr() if mean_rstd is not None else None
    args.colsum = colsum.data_ptr() if colsum is not None else None
    args.tile = tile
    stats = None
    if want_stats:
        parts = lib.tp_linear_stats_parts(ctypes.byref(args))
        assert parts > 0
        stats = torch.full((parts, M, 2), float("nan"), dtype=torch.float32, device=W.device)
        args.row_stats_out = stats.data_ptr()
    if half or t192:
        args.tile = 0
        _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, 3 if t192 else 2)
    try:
        _capi.check(lib.tp_linear(ctypes.byref(args), stream_ptr()), "tp_linear")
    finally:
        if half or t192:
            _capi.set_tuning(_capi.TP_TUNE_GEMM_TILE, 0)
    if sync:
        torch.cuda.synchronize()
    return (C, stats) if want_stats else C


def ln_finalize(stats, ln_dim=1024, eps=1e-6):
    lib = _capi.load_library()
    parts, M, _ = stats.shape
    out = torch.empty(M, 2, dtype=torch.float32, device=stats.device)
    _capi.check(lib.tp_ln_finalize(stats.data_ptr(), parts, M, ln_dim, eps, out.data_ptr(), stream_ptr()),
                "tp_ln_finalize")
    torch.cuda.synchronize()
    return out


def describe_mismatch(got: torch.Tensor, want: torch.Tensor, name: str, tol: float) -> str:
    """Human-readable map of WHERE a 2-D result is wrong (16x16 block granularity) so that a layout
    bug can be diagnosed from one failed GPU run."""
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    if got.dim() > 2:
        got = got.reshape(-1, got.shape[-1])
        want = want.reshape(-1, want.shape[-1])
    err = (got - want).abs()
    scale = want.abs().max().item() + 1e-30
    bad = err > tol * scale
    lines = [f"[{name}] shape={tuple(got.shape)} max|err|={err.max().item():.4e} max|ref|={scale:.4e} "
             f"rel={err.max().item() / scale:.4e} tol={tol:.1e} bad={bad.float().mean().item() * 100:.2f}% "
             f"nan={torch.isnan(got).sum().item()}"]
    if bad.any():
        idx = torch.nonzero(bad)[:8]
        for r, c in idx.tolist():
            lines.append(f"   ({r},{c}): got {got[r, c].item():.6f} want {want[r, c].item():.6f}")
        R, Cc = got.shape
        rb = bad.float().reshape(-1).new_zeros(((R + 15) // 16, (Cc + 15) // 16))
        rows = torch.nonzero(bad)[:, 0] // 16
        cols = torch.nonzero(bad)[:, 1] // 16
        rb.index_put_((rows, cols), torch.ones(rows.shape[0]), accumulate=True)
        lines.append(f"   bad 16x16 blocks: {(rb > 0).sum().item()} of {rb.numel()}; "
                     f"row-blocks hit {sorted(set(rows.tolist()))[:24]} col-blocks hit {sorted(set(cols.tolist()))[:24]}")
        # transposition hint
        if R == Cc:
            lines.append(f"   rel err vs want^T: {((got - want.t()).abs().max() / scale).item():.3e}")
    return "\n".join(lines)


def assert_close(got, want, name, tol):
    got_f = got.detach().float().cpu()
    want_f = want.detach().float().cpu()
    scale = want_f.abs().max().item() + 1e-30
    err = (got_f - want_f).abs().max().item() / scale
    ok = err <= tol and not torch.isnan(got_f).any()
    assert ok, describe_mismatch(got, want, name, tol)
    return err


import contextlib


@contextlib.contextmanager
def training_schedule_for_inference():
    """Inference normally runs schedules the training forward cannot (the backward needs H2, K and V): the fused
    LayerNorm chain (TP_TUNE_FUSE_KV_LN) and, for scale_factor >= 3, the absorbed K/V in-projection
    (TP_TUNE_ABSORB_KV).  Inside this context inference is told to run the training forward's schedule, which makes the
    two bit-identical."""
    _capi.set_tuning(_capi.TP_TUNE_ABSORB_KV, 1)
    _capi.set_tuning(_capi.TP_TUNE_FUSE_KV_LN, 0)
    _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 2)         # the training forward never splits K
    try:
        yield
    finally:
        _capi.set_tuning(_capi.TP_TUNE_ABSORB_KV, 0)
        _capi.set_tuning(_capi.TP_TUNE_FUSE_KV_LN, 1)
        _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 0)


@contextlib.contextmanager
def batch_invariant():
    """TP_TUNE_SPLIT_K = 2: no GEMM of a small batch is split over K, so an image's bits do not depend on the batch it travels
    in (the default gives that up for batches of <= 3 images: 13 % at B = 1; results stay deterministic either way)."""
    _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 2)
    try:
        yield
    finally:
        _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 0)
