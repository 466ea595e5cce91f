"""Triangular statistics (TP_TUNE_TRI_STATS, tp_pack_qr.hip), on a real MI355X: the pack-time Householder factorisation on
its own, and the forward with the centred chain weights + the triangular statistics GEMM against the fp64 oracle and against
the full statistics GEMM it replaces."""
import ctypes

import pytest
import torch

from oracle import tokenpacker_oracle as orc
from tokenpacker_amd import TokenPacker, _capi, synth

pytestmark = pytest.mark.gpu
E = 1024


def _pack_qr(w2, b2):
    lib = _capi.load_test_library()                      # (tp_test_pack_qr: include/tokenpacker_test.h — the same tp_pack_qr.hip kernels the product's pack runs)
    dev = w2.device
    r = torch.full((E, E), float("nan"), dtype=torch.float16, device=dev)
    c = torch.full((E,), float("nan"), dtype=torch.float32, device=dev)
    wbar = torch.full((E + 1,), float("nan"), dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.tp_test_pack_qr_scratch_bytes(), dtype=torch.uint8, device=dev)
    _capi.check(lib.tp_test_pack_qr(w2.data_ptr(), b2.data_ptr() if b2 is not None else None, r.data_ptr(), c.data_ptr(),
                                    wbar.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream),
                "tp_test_pack_qr")
    torch.cuda.synchronize()
    return r, c, wbar


@pytest.mark.parametrize("with_bias,offset", [(True, 0.0), (False, 0.0), (True, 3.0)])
def test_pack_qr_preserves_the_centred_norm(with_bias, offset):
    """|| R h + c~ ||^2 == sum_n ((W2 h + b2)_n - mean)^2 for every h, R upper triangular: the identity the statistics GEMM
    relies on.  `offset` gives W2 and b2 a common shift far larger than their spread (the row's MEAN dominates the layer's
    output — the case the centred form exists for)."""
    g = torch.Generator().manual_seed(4100 + int(with_bias) + int(offset))
    w2 = (torch.randn(E, E, generator=g) * 0.03 + offset * 0.03).to(torch.float16).cuda()
    b2 = (torch.randn(E, generator=g) * 0.1 + offset).float().cuda() if with_bias else None
    r, c, wbar = _pack_qr(w2, b2)
    assert not torch.isnan(r).any() and not torch.isnan(c).any() and not torch.isnan(wbar).any()
    assert float(torch.tril(r.float(), diagonal=-1).abs().max()) == 0.0          # zeros below the diagonal, exactly
    w64 = w2.double()
    b64 = b2.double() if with_bias else torch.zeros(E, dtype=torch.float64, device="cuda")
    assert torch.allclose(wbar[:E].double(), w64.mean(dim=0), rtol=0, atol=1e-7)
    assert abs(float(wbar[E]) - float(b64.mean())) <= 1e-6 * max(1.0, abs(float(b64.mean())))
    h = torch.randn(512, E, generator=g).double().cuda() * 1.5
    y = h @ w64.t() + b64
    want = ((y - y.mean(dim=1, keepdim=True)) ** 2).sum(dim=1)
    got = ((h @ r.double().t() + c.double()) ** 2).sum(dim=1)
    rel = float(((got - want).abs() / want).max())
    print(f"\n[pack-qr] bias={with_bias} offset={offset}: max rel error of the centred sum of squares {rel:.3e}")
    assert rel <= 2e-4            # R is rounded to fp16 (2^-11 per element, random signs over >= 1 elements per row)
    # and the factor is the factor of the centred matrix: R^T R = W2c^T W2c
    w2c = w64 - w64.mean(dim=0, keepdim=True)
    gram = w2c.t() @ w2c
    assert float((r.double().t() @ r.double() - gram).abs().max() / gram.abs().max()) <= 2e-3


def _module(params, s, D, dtype, grid=24):
    m = TokenPacker(raw_grid=grid, hidden_size=D, scale_factor=s)
    m.load_state_dict(params, strict=True)
    return m.to(device="cuda", dtype=dtype).eval().requires_grad_(False)


def _both(m, x, xm):
    out = {}
    try:
        for mode in (0, 1):
            _capi.set_tuning(_capi.TP_TUNE_TRI_STATS, mode)
            with torch.no_grad():
                out[mode] = m((x.cuda(), xm.cuda())).float().cpu()
    finally:
        _capi.set_tuning(_capi.TP_TUNE_TRI_STATS, 0)
    return out


@pytest.mark.parametrize("s", [2, 3, 4])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_forward_triangular_vs_full_statistics(s, dtype):
    D, B = 256, 3
    params = synth.make_params(8100 + s, D)
    x, xm = synth.make_inputs(8200 + s, B, dtype)
    m = _module(params, s, D, dtype)
    m.output_fp32 = True
    out = _both(m, x, xm)
    p_lp = {k: v.to(dtype) for k, v in params.items()}
    y_exact = orc.forward(p_lp, x, xm, scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype)
    e_tri, e_full = orc.rel_err(out[0], y_exact), orc.rel_err(out[1], y_exact)
    print(f"\n[tri-stats] s={s} {dtype}: rel_err triangular {e_tri:.3e} | full statistics {e_full:.3e} | "
          f"between the two {orc.rel_err(out[0], out[1].double()):.3e}")
    gate = 1.0e-3 if s == 2 else 1.1e-3
    assert e_tri <= gate and e_full <= gate
    assert not torch.equal(out[0], out[1])                  # (the knob does select a different computation)


@pytest.mark.parametrize("s", [2, 3])
def test_rows_whose_mean_dominates(s):
    """The second layers' biases get a common offset of 40 standard deviations: H2's row mean dwarfs its spread.  The full
    form subtracts mean x colsum from an accumulator of that size (fp16-rounded chain weights: the rounding error is relative
    to the UNcentred weight); the centred form never sees the mean."""
    dtype, D, B = torch.float16, 256, 2
    params = synth.make_params(8300 + s, D)
    for name in ("k_proj_1.2.bias", "v_proj_1.2.bias"):
        params[name] = params[name] + 40.0 * params[name].std()
    x, xm = synth.make_inputs(8301 + s, B, dtype)
    m = _module(params, s, D, dtype)
    m.output_fp32 = True
    out = _both(m, x, xm)
    p_lp = {k: v.to(dtype) for k, v in params.items()}
    y_exact = orc.forward(p_lp, x, xm, scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype)
    e_tri, e_full = orc.rel_err(out[0], y_exact), orc.rel_err(out[1], y_exact)
    print(f"\n[tri-stats] mean-dominated rows, s={s}: rel_err triangular {e_tri:.3e} | full statistics {e_full:.3e}")
    assert e_tri <= 1.2e-3
    assert e_tri <= e_full * 1.15 + 5e-5


def test_one_image_and_ragged_tail_rows():
    """B = 1 (144 + 576 rows: the 128-tile kernel's triangular K loop, 16 .. 2 K-tiles per column tile) and a batch whose row
    count is not a multiple of the tile."""
    dtype, D = torch.float16, 256
    for B, s, grid in ((1, 2, 24), (1, 3, 24), (5, 2, 6)):
        params = synth.make_params(8400 + B + s, D)
        g = torch.Generator().manual_seed(8401 + B + s)
        x = torch.randn(B, grid * grid, 1024, generator=g).to(dtype)
        xm = torch.randn(B, grid * grid, 4096, generator=g).to(dtype)
        m = _module(params, s, D, dtype, grid)
        m.output_fp32 = True
        out = _both(m, x, xm)
        p_lp = {k: v.to(dtype) for k, v in params.items()}
        y_exact = orc.forward(p_lp, x, xm, scale_factor=s, raw_grid=grid, compute_dtype=torch.float64, io_dtype=dtype)
        e = orc.rel_err(out[0], y_exact)
        print(f"\n[tri-stats] B={B} s={s} grid={grid}: rel_err {e:.3e} (full statistics {orc.rel_err(out[1], y_exact):.3e})")
        assert e <= 1.2e-3
