"""The pair GEMM kernel (tokenpacker_amd/csrc/tp_gemm_pair.hip: two co-resident 4-wave workgroups per CU, 256 x 128 tiles,
three-slot A / W rings, parameters by LDS-DMA) against the kernels it replaces.

It shares the MFMA order and the epilogue code of the other GEMM kernels, so the contract is BIT-IDENTITY: every launch the
pair kernel serves must equal the same launch on the ping-pong / 128-tile kernels bit for bit (which the oracle tests of
test_gpu_kernels.py / test_gpu_forward.py pin to the reference), for every operand form, epilogue and tile-count regime; and
repeated launches must be bit-identical to each other (a race between the two workgroups of a CU, a mis-counted vmcnt or a
ring slot re-targeted too early shows up as run-to-run differences or as a difference from the other kernel).
`tp_debug_counter(TP_COUNTER_PAIR_LAUNCHES)` proves the pair route was actually taken."""
import contextlib

import pytest
import torch

from tokenpacker_amd import _capi, synth
from tests import gpu_util as gu
from tests.test_gpu_forward import _module

pytestmark = pytest.mark.gpu

G, F, S, NS = _capi.TP_LINEAR_GELU, _capi.TP_LINEAR_LN_FOLD, _capi.TP_LINEAR_ROW_STATS, _capi.TP_LINEAR_NO_STORE


@contextlib.contextmanager
def pair(mode, stagger=None):
    _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, mode)
    if stagger is not None:
        _capi.set_tuning(_capi.TP_TUNE_PAIR_STAGGER, stagger)
    try:
        yield
    finally:
        _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_STAGGER, 100)


def launches():
    return _capi.load_library().tp_debug_counter(_capi.TP_COUNTER_PAIR_LAUNCHES)


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(dtype)


def _both(fn):
    """fn() under the pair kernel (forced wherever supported) and with it off; asserts the pair route was taken."""
    with pair(1):
        ref = fn()
    n0 = launches()
    with pair(2):
        got = fn()
        again = fn()
    assert launches() > n0, "the launch did not take the pair route"
    return got, again, ref


def _eq(a, b, what):
    if isinstance(a, tuple):
        for x, y in zip(a, b):
            _eq(x, y, what)
        return
    assert torch.equal(a, b), gu.describe_mismatch(a, b, what, 0.0)


# M: whole tiles, a ragged last tile (rows past M dropped), fewer rows than one tile; N: one column tile .. 32; K: 3 K-tiles (the
# minimum: prologue + both tail forms back to back), 4, 16, 64
SHAPES = [(256, 128, 192), (512, 256, 256), (1000, 1024, 1024), (300, 128, 4096), (77, 256, 1024), (4096, 512, 320),
          (147456 // 8, 2048, 4096), (36864 // 4, 4096, 1024)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_pair_linear_plain_bias_gelu(dtype, M, N, K):
    A = _rand((M, K), dtype, 1)
    W = _rand((N, K), dtype, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, 3)
    for out_dtype in (torch.float16, torch.bfloat16, torch.float32):
        for flags, b in ((0, None), (0, bias), (G, bias)):
            got, again, ref = _both(lambda: gu.linear(A, W, bias=b, flags=flags, out_dtype=out_dtype))
            _eq(got, ref, f"pair vs others {dtype} -> {out_dtype} flags {flags} {M}x{N}x{K}")
            _eq(again, got, "pair run-to-run")


@pytest.mark.parametrize("M,N,K", [(1000, 1024, 1024), (512, 128, 256), (9000, 1024, 1024)])
def test_pair_linear_ln_fold_and_row_stats(M, N, K):
    dtype = torch.float16
    A = _rand((M, K), dtype, 4)
    W = _rand((N, K), dtype, 5, K ** -0.5)
    bias = _rand((N,), torch.float32, 6)
    colsum = _rand((N,), torch.float32, 7)
    mr = torch.rand(M, 2, device="cuda") + 0.5
    # LayerNorm fold (reads per-row (mean, rstd) + colsum through the DMA-staged parameters)
    got, again, ref = _both(lambda: gu.linear(A, W, bias=bias, flags=F, mean_rstd=mr, colsum=colsum))
    _eq(got, ref, "LN fold"); _eq(again, got, "LN fold run-to-run")
    # row statistics of the rounded output (the reduction scratch lives in W-ring slot 2)
    got, again, ref = _both(lambda: gu.linear(A, W, bias=bias, want_stats=True))
    _eq(got, ref, "row stats"); _eq(again, got, "row stats run-to-run")
    # statistics only
    got, again, ref = _both(lambda: gu.linear(A, W, bias=bias, flags=NS, want_stats=True)[1])
    _eq(got, ref, "stats only"); _eq(again, got, "stats only run-to-run")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_pair_linear_strided_rows(dtype):
    """The tower's [:, 1:] slices: rows in batches of 576 with a batch stride of 577 rows (AMODE 1)."""
    B, K, N = 9, 4096, 2048
    buf = _rand((B, 577, K), dtype, 8)
    W = _rand((N, K), dtype, 9, K ** -0.5)
    bias = _rand((N,), torch.float32, 10)
    A = buf[:, 1:]
    fn = lambda: gu.linear(A, W, bias=bias, flags=G, out_dtype=torch.float16, rows_per_batch=576,
                           a_batch_stride=577 * K, lda=K, M=B * 576)
    got, again, ref = _both(fn)
    _eq(got, ref, "strided A"); _eq(again, got, "strided A run-to-run")
    _eq(got, gu.linear(A.contiguous().reshape(B * 576, K), W, bias=bias, flags=G, out_dtype=torch.float16), "strided vs contiguous")


def _forward_cases():
    # (scale_factor, hidden_size, batch, dtype, layout)
    return [(2, 256, 9, torch.bfloat16, "tower"), (2, 256, 40, torch.float16, "contiguous"), (3, 256, 9, torch.bfloat16, "tower"),
            (4, 256, 5, torch.float16, "tower"), (2, 4096, 4, torch.bfloat16, "tower"), (2, 5120, 2, torch.bfloat16, "contiguous")]


@pytest.mark.parametrize("s,D,B,dtype,layout", _forward_cases())
def test_forward_on_the_pair_kernel_is_the_same_function(s, D, B, dtype, layout):
    """The whole path with every supported launch forced onto the pair kernel: region-major strided first layer, triangular
    statistics, attention inside the K / V launches' epilogues, acc_init chains, the absorbed schedule's grouped GEMMs."""
    m = _module(synth.make_params(300 + s, D), s, D, dtype)
    x, xm = synth.make_inputs(31, B, dtype, layout)
    if layout == "tower":
        xg = torch.zeros(B, 577, 1024, dtype=dtype, device="cuda"); xmg = torch.zeros(B, 577, 4096, dtype=dtype, device="cuda")
        xg[:, 1:] = x.cuda(); xmg[:, 1:] = xm.cuda()
        x, xm = xg[:, 1:], xmg[:, 1:]
    else:
        x, xm = x.cuda(), xm.cuda()
    with torch.no_grad(), gu.batch_invariant():
        got, again, ref = _both(lambda: m((x, xm)))
        _eq(got, ref, f"forward s={s} D={D} B={B}")
        _eq(again, got, "forward run-to-run")
        # the hidden states as four parts (K split over four tensors: AMODE 2)
        if layout == "tower":
            parts = tuple(xm[..., i * 1024:(i + 1) * 1024] for i in range(4))
            gp, _, rp = _both(lambda: m((x, parts)))
            _eq(gp, got, "four-part first layer on the pair kernel"); _eq(rp, ref, "four-part first layer")
        # fp32 output (mlp[2] writes fp32)
        m.output_fp32 = True
        g32, _, r32 = _both(lambda: m((x, xm)))
        m.output_fp32 = False
        _eq(g32, r32, "fp32 output")


def test_full_size_on_the_pair_kernel_is_race_free():
    """BASELINE config 2 (B = 256, s = 2, D = 4096, bf16) with every launch forced onto the pair kernel: equal to the pair-less path
    bit for bit, 64 repeats of 4 images give 64 identical copies, and ten forwards under different stagger settings (which move
    the two workgroups of a CU against each other) are all the same bits.  The DEFAULT policy sends only the short-K launches of
    1.5 .. 2.5 rounds there (the query-side GEMMs at this batch): also the same bits."""
    dtype, D, s, B = torch.bfloat16, 4096, 2, 256
    m = _module(synth.make_params(6, D), s, D, dtype)
    x4, xm4 = synth.make_inputs(10, 4, dtype)
    x = x4.repeat(B // 4, 1, 1).cuda()
    xm = xm4.repeat(B // 4, 1, 1).cuda()
    with torch.no_grad():
        with pair(1):
            ref = m((x, xm))
        n0 = launches()
        y0 = m((x, xm))
        assert launches() - n0 >= 1, "the default policy should route the query-side GEMMs of a B = 256 forward to the pair kernel"
        _eq(y0, ref, "default policy vs pair off at B = 256")
        n0 = launches()
        with pair(2):
            y = m((x, xm))
        assert launches() - n0 >= 8
        _eq(y, ref, "pair forced vs pair off at B = 256")
        yr = y.reshape(B // 4, 4, 144, D)
        assert torch.equal(yr, yr[:1].expand_as(yr)), "batch elements must not interact"
        for st in (0, 100, 37, 250, 100, 0, 63, 100, 180, 100):
            with pair(2, st):
                _eq(m((x, xm)), y, f"stagger {st}")
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [24, 32, 36, 100])
def test_mid_batches_both_policies(B):
    """Batches whose launches straddle the pair kernel's tile-count threshold: default policy, forced pair, pair off."""
    dtype, D, s = torch.bfloat16, 4096, 2
    m = _module(synth.make_params(7, D), s, D, dtype)
    x, xm = synth.make_inputs(11, 4, dtype)
    x = x.repeat((B + 3) // 4, 1, 1)[:B].cuda(); xm = xm.repeat((B + 3) // 4, 1, 1)[:B].cuda()
    with torch.no_grad():
        got, again, ref = _both(lambda: m((x, xm)))
        _eq(got, ref, f"B={B}"); _eq(again, got, "run-to-run")
        _eq(m((x, xm)), ref, f"default policy B={B}")
