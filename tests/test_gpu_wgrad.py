"""tp_wgrad: dW = dY^T · X read from the row-major activations (the ping-pong GEMM with K-major operands and
transposing LDS reads) against torch.matmul in fp64 on the same 16-bit inputs.  Through the C ABI."""
import ctypes

import pytest
import torch

from tokenpacker_amd import _capi

pytestmark = pytest.mark.gpu

DT = {torch.bfloat16: _capi.TP_BF16, torch.float16: _capi.TP_F16, torch.float32: _capi.TP_F32}


def _wgrad(dy, x, rows, n_out, k_in, out_dtype, rpb=0, bstride=0, flags=0):
    lib = _capi.load_library()
    ws = torch.empty(lib.tp_wgrad_workspace_bytes(n_out, k_in), dtype=torch.uint8, device="cuda")
    dw = torch.full((n_out, k_in), float("nan"), dtype=out_dtype, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.tp_wgrad(dy.data_ptr(), dy.stride(-2), x.data_ptr(), x.stride(-2), rpb, bstride, rows, n_out, k_in, DT[dy.dtype],
                      dw.data_ptr(), DT[out_dtype], flags, ws.data_ptr(), ws.numel(), st)
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return dw


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,n_out,k_in", [(64, 256, 256), (100, 256, 256), (1000, 512, 256), (4096, 1024, 1024),
                                              (2304, 264, 512), (18432, 2048, 4096)])
def test_wgrad_vs_fp64(dtype, rows, n_out, k_in):
    g = torch.Generator(device="cuda").manual_seed(rows + n_out)
    dy = torch.randn(rows, n_out, device="cuda", generator=g).to(dtype)
    x = torch.randn(rows, k_in, device="cuda", generator=g).to(dtype)
    ref = dy.double().t() @ x.double()
    for out_dtype in (torch.float32, dtype):
        dw = _wgrad(dy, x, rows, n_out, k_in, out_dtype)
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        tol = 1e-5 if out_dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
        assert err <= tol, (rows, n_out, k_in, out_dtype, err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,n_out,k_in", [(100, 256, 256), (1000, 512, 256), (4608, 1024, 1024), (36864, 4096, 1024)])
def test_wgrad_with_transposed_activations(dtype, rows, n_out, k_in):
    """dy read in place (K-major), x given as X^T [k_in, rpad] (K-contiguous, zero-padded): both layouts meet in one MFMA."""
    g = torch.Generator(device="cuda").manual_seed(rows + k_in)
    dy = torch.randn(rows, n_out, device="cuda", generator=g).to(dtype)
    x = torch.randn(rows, k_in, device="cuda", generator=g).to(dtype)
    rpad = (rows + 1023) // 1024 * 1024
    xt = torch.zeros(k_in, rpad, dtype=dtype, device="cuda")
    xt[:, :rows] = x.t()
    ref = dy.double().t() @ x.double()
    dw = _wgrad(dy, xt, rows, n_out, k_in, torch.float32, flags=_capi.TP_WGRAD_X_TRANSPOSED)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err <= 1e-5, err


def test_wgrad_strided_and_batched_sources():
    """dy as a column slice of a wider matrix (ldy > n_out), x as the tower's [:, 1:] slices (batch-strided rows)."""
    dtype, B, T, n_out, k_in = torch.bfloat16, 5, 576, 512, 1024
    g = torch.Generator(device="cuda").manual_seed(7)
    dy_wide = torch.randn(B * T, 2 * n_out, device="cuda", generator=g).to(dtype)
    dy = dy_wide[:, n_out:]
    hidden = torch.randn(B, T + 1, k_in, device="cuda", generator=g).to(dtype)
    x = hidden[:, 1:]
    ref = dy.double().t() @ x.reshape(B * T, k_in).double()
    dw = _wgrad(dy, x, B * T, n_out, k_in, torch.float32, rpb=T, bstride=hidden.stride(0))
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err <= 1e-5, err


def test_wgrad_is_deterministic_and_rejects_bad_shapes():
    lib = _capi.load_library()
    dy = torch.randn(3000, 512, device="cuda").to(torch.bfloat16)
    x = torch.randn(3000, 768, device="cuda").to(torch.bfloat16)
    a = _wgrad(dy, x, 3000, 512, 768, torch.float32)
    b = _wgrad(dy, x, 3000, 512, 768, torch.float32)
    assert torch.equal(a, b)
    ws = torch.empty(lib.tp_wgrad_workspace_bytes(512, 200), dtype=torch.uint8, device="cuda")
    rc = lib.tp_wgrad(dy.data_ptr(), 512, x.data_ptr(), 768, 0, 0, 3000, 512, 200, _capi.TP_BF16, a.data_ptr(), _capi.TP_F32,
                      0, ws.data_ptr(), ws.numel(), None)
    assert rc == _capi.TP_ERR_INVALID_ARG and "256" in _capi.last_error()
