"""Per-kernel parity on a real MI355X, through the C ABI (tp_linear / tp_point_queries /
tp_region_attention).  Reference math is the oracle (fp64 on the same rounded operands)."""
import ctypes
import math

import pytest
import torch

from oracle import tokenpacker_oracle as orc
from tokenpacker_amd import _capi, synth
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
# fp32-out results differ from fp64 math only by fp32 accumulation order
TOL_F32OUT = 2e-5
# one rounding of the output to T: half-ulp relative to the value, measured against max|ref|
TOL_ROUND = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


# block-tile variants of the MFMA kernels: 128 (two-phase, tp_gemm.hip), 256 (ping-pong 8-wave, tp_gemm8.hip)
TILES = [128, 256]
# the ping-pong kernel's tile shapes: 256 x 256 | every tile a 128 x 256 half tile (TP_TUNE_GEMM_TILE = 2)
HALVES = [False, True]


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


def _ref_linear(A, W, bias=None):
    y = A.double().cpu() @ W.double().cpu().t()
    return y if bias is None else y + bias.double().cpu()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 512, 256), (300, 256, 128), (1000, 1024, 1024), (77, 128, 4096)])
def test_linear_plain(dtype, tile, M, N, K):
    if abs(tile) == 256 and N % 256:
        pytest.skip("tile 256 needs N % 256 == 0")
    A = _rand((M, K), dtype, 1)
    W = _rand((N, K), dtype, 2, K ** -0.5)      # asymmetric, non-square data: catches transposes
    ref = _ref_linear(A, W)
    C32 = gu.linear(A, W, out_dtype=torch.float32, tile=tile)
    gu.assert_close(C32, ref, f"linear f32out {dtype} tile{tile} {M}x{N}x{K}", TOL_F32OUT)
    C = gu.linear(A, W, tile=tile)
    assert C.dtype == dtype
    gu.assert_close(C, ref, f"linear {dtype} tile{tile} {M}x{N}x{K}", TOL_ROUND[dtype] * 1.01 + TOL_F32OUT)
    # the T output must be exactly the rounding of the fp32 output (same accumulators)
    assert torch.equal(C, C32.to(dtype))
    # mixed output type (the first layer of a bf16 model writes fp16 activations)
    other = torch.float16 if dtype == torch.bfloat16 else torch.bfloat16
    Co = gu.linear(A, W, tile=tile, out_dtype=other)
    assert torch.equal(Co, C32.to(other))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", TILES)
def test_linear_bias_gelu(dtype, tile):
    M, N, K = 640, 512, 256
    A = _rand((M, K), dtype, 3)
    W = _rand((N, K), dtype, 4, K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    ref = orc.gelu_erf(_ref_linear(A, W, bias))
    C32 = gu.linear(A, W, bias=bias, flags=_capi.TP_LINEAR_GELU, tile=tile, out_dtype=torch.float32)
    gu.assert_close(C32, ref, f"linear+bias+gelu {dtype} tile{tile}", 3e-5)     # erf approx: <= 1.5e-7 abs


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", TILES)
def test_linear_strided_batch_rows(dtype, tile):
    """A given as the tower's [:,1:] slice of a CLS-prefixed [B,577,K] buffer (clip_encoder.py:37-38)."""
    B, T, K, N = 3, 576, 256, 256
    buf = _rand((B, T + 1, K), dtype, 6)
    A = buf[:, 1:]
    assert not A.is_contiguous()
    W = _rand((N, K), dtype, 7, K ** -0.5)
    ref = _ref_linear(A.reshape(B * T, K), W)
    C32 = gu.linear(A, W, out_dtype=torch.float32, tile=tile, M=B * T, rows_per_batch=T,
                    a_batch_stride=A.stride(0), lda=A.stride(1))
    gu.assert_close(C32, ref, f"linear strided {dtype} tile{tile}", TOL_F32OUT)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", TILES)
def test_linear_row_stats_and_ln_fold(dtype, tile):
    """GEMM#1 emits per-row, per-128-column (mean, M2) partials of its ROUNDED output; GEMM#2 consumes them to apply
    LayerNorm folded into its epilogue.  Reference: LN then linear in fp64 (builder.py:112,120 + in-proj)."""
    M, E = 700, 1024
    A = _rand((M, 256), dtype, 8)
    W1 = _rand((E, 256), dtype, 9, 256 ** -0.5)
    b1 = (0.3 * torch.randn(E, generator=torch.Generator().manual_seed(10)) + 0.2).cuda()   # non-zero mean
    H, stats = gu.linear(A, W1, bias=b1, tile=tile, want_stats=True)
    parts = stats.shape[0]
    assert parts == E // 128          # one slab per 128 columns, whatever the tile
    Hd = H.double().cpu()
    slabs = Hd.reshape(M, parts, 128)
    st = stats.double().cpu()                                    # [parts][M][2]
    assert torch.allclose(st[:, :, 0].t(), slabs.mean(2), rtol=0, atol=2e-6), "slab means"
    m2 = ((slabs - slabs.mean(2, keepdim=True)) ** 2).sum(2)
    assert torch.allclose(st[:, :, 1].t(), m2, rtol=2e-5, atol=1e-5), "slab sums of squared deviations"

    # LN-fold operands prepared exactly like tp_pack_weights does
    g = torch.Generator().manual_seed(11)
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(dtype)
    beta = (0.1 * torch.randn(E, generator=g)).to(dtype)
    W2 = (torch.randn(E, E, generator=g) * E ** -0.5).to(dtype)
    b2 = (0.1 * torch.randn(E, generator=g)).to(dtype)
    W2p = (W2.float() * gamma.float()).to(dtype)
    colsum = W2p.float().sum(1).cuda()
    biasp = (W2.float() @ beta.float() + b2.float()).cuda()
    ref = orc.linear(orc.layer_norm(Hd, gamma.double(), beta.double()), W2.double(), b2.double())
    mr = gu.ln_finalize(stats, E, 1e-6)
    mu_ref, var_ref = Hd.mean(1), Hd.var(1, unbiased=False)
    assert torch.allclose(mr[:, 0].double().cpu(), mu_ref, atol=1e-5)
    assert torch.allclose(mr[:, 1].double().cpu(), 1.0 / torch.sqrt(var_ref + 1e-6), rtol=1e-4)
    C32 = gu.linear(H, W2p.cuda(), bias=biasp, flags=_capi.TP_LINEAR_LN_FOLD, tile=tile, out_dtype=torch.float32,
                    mean_rstd=mr, colsum=colsum)
    # differences: W·gamma rounded to T once (pack) vs exact gamma in the reference
    gu.assert_close(C32, ref, f"ln-fold {dtype} tile{tile}", TOL_ROUND[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_row_stats_do_not_depend_on_tile(dtype):
    A = _rand((512, 256), dtype, 12)
    W = _rand((1024, 256), dtype, 13, 256 ** -0.5)
    H1, s1 = gu.linear(A, W, tile=128, want_stats=True)
    H2, s2 = gu.linear(A, W, tile=256, want_stats=True)
    H3, s3 = gu.linear(A, W, tile=256, half=True, want_stats=True)
    assert torch.equal(H1, H2) and torch.equal(s1, s2)
    assert torch.equal(H1, H3) and torch.equal(s1, s3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 1024])
@pytest.mark.parametrize("half", HALVES)
def test_pingpong_short_and_odd_k(dtype, K, half):
    """K-tile counts 1..5 and 16 walk every prologue / tail variant of the ping-pong main loop."""
    M, N = 700, 512
    A = _rand((M, K), dtype, 50 + K)
    W = _rand((N, K), dtype, 51 + K, K ** -0.5)
    ref = _ref_linear(A, W)
    if half and K < 128:
        pytest.skip("half tiles need K >= 128")
    C32 = gu.linear(A, W, out_dtype=torch.float32, tile=256, half=half)
    gu.assert_close(C32, ref, f"pingpong K={K} {dtype} half={half}", TOL_F32OUT)
    assert torch.equal(C32, gu.linear(A, W, out_dtype=torch.float32, tile=128))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(700, 512, 64), (700, 512, 128), (192, 256, 192), (1000, 1024, 320), (18432, 2048, 4096), (4608, 4096, 1024),
                                   (4609, 512, 1024), (191, 256, 256)])
def test_pingpong_192_row_tiles(dtype, M, N, K):
    """192 x 256 tiles (tp_gemm8.hip T192: a short a1 quadrant, one-instruction G3 group, its own counted waits) against the 128-tile
    kernel, bit for bit: every prologue / tail form (1 .. 64 K-tiles), whole and ragged last tiles, bias + GELU, every output dtype,
    repeated launches (a mis-counted vmcnt or a wrongly mapped G3 row shows up as a differing tile)."""
    A = _rand((M, K), dtype, 60 + K)
    W = _rand((N, K), dtype, 61 + K, K ** -0.5)
    bias = _rand((N,), torch.float32, 62)
    for out_dtype in (torch.float16, torch.bfloat16, torch.float32):
        for flags, b in ((0, None), (_capi.TP_LINEAR_GELU, bias)):
            ref = gu.linear(A, W, bias=b, flags=flags, out_dtype=out_dtype, tile=128)
            for rep in range(2):
                got = gu.linear(A, W, bias=b, flags=flags, out_dtype=out_dtype, t192=True)
                assert torch.equal(got, ref), gu.describe_mismatch(got, ref, f"192-row tiles {dtype}->{out_dtype} {M}x{N}x{K} flags {flags} rep {rep}", 0.0)


def test_pingpong_192_row_tiles_strided_region_major_a():
    """The first K/V layer's operand form on 192-row tiles: rows in per-image batches with a batch stride (the tower's [:, 1:] slices)."""
    dtype, B, T, K, N = torch.bfloat16, 32, 576, 4096, 2048
    full = _rand((B, T + 1, K), dtype, 70)
    A = full[:, 1:, :]
    W = _rand((N, K), dtype, 71, K ** -0.5)
    bias = _rand((N,), torch.float32, 72)
    kw = dict(bias=bias, flags=_capi.TP_LINEAR_GELU, out_dtype=torch.float16, rows_per_batch=T, a_batch_stride=A.stride(0), lda=A.stride(1), M=B * T)
    ref = gu.linear(A, W, tile=128, **kw)
    got = gu.linear(A, W, t192=True, **kw)
    assert torch.equal(got, ref), gu.describe_mismatch(got, ref, "192-row tiles, strided A", 0.0)
    auto = gu.linear(A, W, **kw)                       # (the default route of a 32-image shard's first layer)
    assert torch.equal(auto, ref)


@pytest.mark.parametrize("M,N,K,flags", [(36864, 4096, 4096, 0), (147456, 1024, 1024, _capi.TP_LINEAR_ROW_STATS),
                                         (36864, 4096, 1024, _capi.TP_LINEAR_GELU), (20000, 2048, 4096, _capi.TP_LINEAR_GELU)])
@pytest.mark.parametrize("half", HALVES)
def test_pingpong_full_size_race_screen(M, N, K, flags, half):
    """Full-size shapes of the B=256 path, random data, every CU busy for many rounds: the
    ping-pong kernel must reproduce the two-phase 128-tile kernel (tp_gemm.hip: other tiles, other main loop, same epilogue) BIT
    FOR BIT (both accumulate each K-slab in the same order), on every one of several back-to-back launches — a DMA/LDS race shows up as a differing tile."""
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(77)
    A = torch.randn(M, K, generator=g, device="cuda").to(dtype)
    W = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(dtype)
    bias = torch.randn(N, generator=g, device="cuda")
    want_stats = bool(flags & _capi.TP_LINEAR_ROW_STATS)
    fl = flags & ~_capi.TP_LINEAR_ROW_STATS
    ref = gu.linear(A, W, bias=bias, flags=fl, tile=128, want_stats=want_stats, out_dtype=torch.float16)
    for rep in range(4):
        got = gu.linear(A, W, bias=bias, flags=fl, tile=256, half=half, want_stats=want_stats, out_dtype=torch.float16,
                        sync=False)
        if want_stats:
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), f"launch {rep}"
        else:
            assert torch.equal(got, ref), f"launch {rep}: {(got.float() - ref.float()).abs().max().item()}"
    # and the two-phase kernel itself against fp32 math on a row sample (transposition-detecting: N != K or random)
    rows = torch.randint(0, M, (64,), generator=torch.Generator().manual_seed(1)).cuda()
    y = A[rows].float() @ W.float().t() + bias
    if fl & _capi.TP_LINEAR_GELU:
        y = torch.nn.functional.gelu(y)
    c = (ref[0] if want_stats else ref)[rows].float()
    assert (c - y).abs().max() <= 2e-3 * y.abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("s", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("layout", ["contiguous", "tower"])
def test_point_queries_bit_exact(dtype, s, layout):
    B = 3
    x, _ = synth.make_inputs(21, B, dtype, layout)
    xg = x.cuda() if layout == "contiguous" else None
    if layout == "tower":
        buf = torch.zeros(B, 577, 1024, dtype=dtype).cuda()
        buf[:, 1:] = x.cuda()
        xg = buf[:, 1:]
    M = (24 // s) ** 2
    q0 = torch.empty(B, M, 1024, dtype=torch.float16, device="cuda")     # activations are fp16
    lib = _capi.load_library()
    desc = _capi.make_desc(B, 24, s, 4096, gu.DT[dtype])
    _capi.check(lib.tp_point_queries(ctypes.byref(desc), xg.data_ptr(), _capi.strides3(xg.stride()),
                                     q0.data_ptr(), gu.stream_ptr()), "tp_point_queries")
    torch.cuda.synchronize()
    want = orc.point_queries(x.float(), 24, s, io_dtype=dtype)     # fp32 math, ONE rounding to the input dtype
    got = q0.float().cpu()
    # bit-exact wherever the value is a normal fp16 number; bf16 values below 2^-14 land on fp16's
    # subnormal grid (quantum 2^-24 = 6e-8 absolute)
    normal = want.abs() >= 2.0 ** -14
    assert torch.equal(got[normal], want[normal]), gu.describe_mismatch(q0, want, f"point_queries s={s}", 0.0)
    assert (got - want).abs().max() <= 2.0 ** -25


@pytest.mark.parametrize("dtype", [torch.float16])      # activations between kernels are always fp16
@pytest.mark.parametrize("s", [1, 2, 3, 4, 6, 12])
def test_region_attention(dtype, s):
    B, g, E, H = 2, 24, 1024, 8
    G = g // s
    M = G * G
    q = _rand((B, M, E), dtype, 31)
    k = _rand((B, g * g, E), dtype, 32)
    v = _rand((B, g * g, E), dtype, 33)
    o = torch.empty(B, M, E, dtype=dtype, device="cuda")
    lib = _capi.load_library()
    desc = _capi.make_desc(B, g, s, 4096, gu.DT[dtype])
    _capi.check(lib.tp_region_attention(ctypes.byref(desc), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(),
                                        gu.stream_ptr()), "tp_region_attention")
    torch.cuda.synchronize()
    d = E // H
    Q = q.double().cpu().reshape(B, G, G, H, d) / math.sqrt(d)
    K = orc.region_gather(k.double().cpu(), g, s).reshape(B, G, G, s * s, H, d)
    V = orc.region_gather(v.double().cpu(), g, s).reshape(B, G, G, s * s, H, d)
    P = torch.softmax(torch.einsum("bijhd,bijkhd->bijhk", Q, K), dim=-1)
    ref = torch.einsum("bijhk,bijkhd->bijhd", P, V).reshape(B, M, E)
    gu.assert_close(o, ref, f"region_attention {dtype} s={s}", TOL_ROUND[dtype] * 1.05 + 1e-5)


def test_region_attention_peaked_softmax():
    """Large logits (|q·k|/sqrt(d) ~ 60): exercises the running-max rescale across key groups."""
    dtype, s, B, g, E = torch.float16, 4, 1, 24, 1024
    M = (g // s) ** 2
    q = _rand((B, M, E), dtype, 41, 6.0)
    k = _rand((B, g * g, E), dtype, 42, 1.0)
    v = _rand((B, g * g, E), dtype, 43)
    o = torch.empty(B, M, E, dtype=dtype, device="cuda")
    lib = _capi.load_library()
    desc = _capi.make_desc(B, g, s, 4096, gu.DT[dtype])
    _capi.check(lib.tp_region_attention(ctypes.byref(desc), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(),
                                        gu.stream_ptr()), "tp_region_attention")
    torch.cuda.synchronize()
    G, H, d = g // s, 8, 128
    Q = q.double().cpu().reshape(B, G, G, H, d) / math.sqrt(d)
    K = orc.region_gather(k.double().cpu(), g, s).reshape(B, G, G, s * s, H, d)
    V = orc.region_gather(v.double().cpu(), g, s).reshape(B, G, G, s * s, H, d)
    P = torch.softmax(torch.einsum("bijhd,bijkhd->bijhk", Q, K), dim=-1)
    assert P.max() > 0.99          # the case really is peaked
    ref = torch.einsum("bijhk,bijkhd->bijhd", P, V).reshape(B, M, E)
    gu.assert_close(o, ref, "region_attention peaked", 2.0 ** -11 * 1.05 + 1e-5)


@pytest.mark.parametrize("s", [2, 3, 4, 6, 8])
def test_region_attention_absorbed(s):
    """The attention of the absorbed schedule on its own (tp_region_attention_absorbed): rows of H2 normalised on load
    with their (mean, rstd), per-head logits against qt, softmax over the region, u_h = sum_t p_t n^v_t — against the
    same sums in fp64.  Rows carry a large common offset so that normalise-on-load is really exercised."""
    B, g, E, H = 2, 24, 1024, 8
    G = g // s
    M, N = G * G, g * g
    gen = torch.Generator().manual_seed(50 + s)
    qt = (0.4 * torch.randn(B, M, H, E, generator=gen)).half().cuda()
    h2 = [(1.7 * torch.randn(B, N, E, generator=gen) + 3.0 * torch.randn(B, N, 1, generator=gen)).half().cuda() for _ in range(2)]
    mr = []
    for h in h2:
        hd = h.double()
        mu, var = hd.mean(-1), hd.var(-1, unbiased=False)
        mr.append(torch.stack([mu, 1.0 / torch.sqrt(var + 1e-6)], dim=-1).float().reshape(B * N, 2).contiguous())
    u = torch.empty(B, M, H, E, dtype=torch.float16, device="cuda")
    lib = _capi.load_library()
    desc = _capi.make_desc(B, g, s, 4096, _capi.TP_F16)
    _capi.check(lib.tp_region_attention_absorbed(ctypes.byref(desc), qt.data_ptr(), h2[0].data_ptr(), h2[1].data_ptr(),
                                                 mr[0].data_ptr(), mr[1].data_ptr(), u.data_ptr(), gu.stream_ptr()),
                "tp_region_attention_absorbed")
    torch.cuda.synchronize()
    n = [((h.double().cpu() - m_[:, 0].double().cpu().reshape(B, N, 1)) * m_[:, 1].double().cpu().reshape(B, N, 1))
         for h, m_ in zip(h2, mr)]
    nk = orc.region_gather(n[0], g, s)                      # [B, G, G, s*s, E]
    nv = orc.region_gather(n[1], g, s)
    q = qt.double().cpu().reshape(B, G, G, H, E)
    P = torch.softmax(torch.einsum("bijhe,bijke->bijhk", q, nk) / math.sqrt(128), dim=-1)
    ref = torch.einsum("bijhk,bijke->bijhe", P, nv).reshape(B, M, H, E)
    gu.assert_close(u.reshape(B * M * H, E), ref.reshape(B * M * H, E), f"region_attention_absorbed s={s}", 2.0 ** -11 * 1.05 + 1e-5)
