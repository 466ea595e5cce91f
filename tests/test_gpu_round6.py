"""Round-6 additions on a real MI355X: the decoupled K launch (TP_TUNE_DECOUPLE_K: the K launch of the s = 2 schedule writes raw
logits and no longer depends on the K/V row statistics; the V launch applies the K rows' rstd) — an A/B that measured null and
ships OFF, kept parity-tested: against the fp64 oracle at every batch regime (128-tile kernel with in-kernel LayerNorm merge,
half tiles, full tiles), both placements bit-identical to each other, and batch-invariant."""
import pytest
import torch

from oracle import tokenpacker_oracle as orc
from tokenpacker_amd import _capi, synth
from tests.test_gpu_forward import _module

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B", [1, 9, 48])
def test_decoupled_k_launch_matches_the_oracle_on_either_stream(dtype, B):
    D, s = 256, 2
    params = synth.make_params(611, D)
    x, xm = synth.make_inputs(612, B, dtype)
    p_lp = {k: v.to(dtype) for k, v in params.items()}
    y_exact = orc.forward(p_lp, x[:4], xm[:4], scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype)
    xg, xmg = x.cuda(), xm.cuda()
    outs = {}
    for mode in (0, 1, 2):
        m = _module(params, s, D, dtype)
        m.output_fp32 = True
        m.tuning = _capi.TuningContext(decouple_k=mode)
        with torch.no_grad():
            outs[mode] = m((xg, xmg))
            again = m((xg, xmg))
        torch.cuda.synchronize()
        assert torch.equal(outs[mode], again)
        e = orc.rel_err(outs[mode][:4], y_exact[:B])
        print(f"\n[parity] DECOUPLE_K={mode} B={B} {dtype}: rel_err={e:.3e}")
        assert e <= 1e-3
    assert torch.equal(outs[1], outs[2]), "the placement of the K launch must not change a bit"
    assert not torch.equal(outs[0], outs[1]) or B == 0      # (a different rounding order: the knob really selects another form)
    # batch invariance of the raw-logit form: image k of the batch == the same image alone (SPLIT_K = 2: no K-split of tiny batches)
    if B > 1:
        m = _module(params, s, D, dtype)
        m.output_fp32 = True
        m.tuning = _capi.TuningContext(decouple_k=1, split_k=2)
        with torch.no_grad():
            y_all = m((xg, xmg))
            for k in (0, B - 1):
                assert torch.equal(m((xg[k:k + 1], xmg[k:k + 1])), y_all[k:k + 1])


def test_decoupled_k_launch_needs_the_centred_chain():
    """With TP_TUNE_TRI_STATS = 1 (means are not zero) the knob falls back to the round-5 form: same bits as DECOUPLE_K = 0."""
    D, s, B, dtype = 256, 2, 5, torch.float16
    params = synth.make_params(613, D)
    x, xm = synth.make_inputs(614, B, dtype)
    ys = []
    for mode in (0, 1):
        m = _module(params, s, D, dtype)
        m.tuning = _capi.TuningContext(decouple_k=mode, tri_stats=1)
        with torch.no_grad():
            ys.append(m((x.cuda(), xm.cuda())))
    assert torch.equal(ys[0], ys[1])


@pytest.mark.parametrize("M", [200, 4096, 65280, 65536, 65700])
def test_a_k_dup_gemm_is_bit_identical_on_both_routes(M):
    """the absorbed schedule's per-head V GEMM (tp_linear_args.a_k_dup: every K-tile of u serves the hi_t | lo_t K-tile
    pair of the weight) picks its kernel by launch size — the pair kernel from half a round of its tiles on (M >= 65536 rows at
    N = 128), the 128-tile kernel below — so an image's bits depend on the batch unless both kernels agree bit for bit.  Direct test
    on both routes (TP_TUNE_PAIR_GEMM 1 = never / 2 = wherever supported / 0 = by size), ragged M and M around the threshold, and
    against an fp32 reference of the same contraction."""
    import ctypes
    from tests.gpu_util import DT, stream_ptr
    lib = _capi.load_library()
    E, N = 1024, 128
    g = torch.Generator(device="cuda").manual_seed(700 + M)
    A = torch.randn(M, E, device="cuda", generator=g).to(torch.float16)
    Wp = (torch.randn(N, 2 * E, device="cuda", generator=g) * E ** -0.5).to(torch.float16)      # rows = K-tile pairs [hi_0 lo_0 hi_1 lo_1 ..]
    bias = torch.randn(N, device="cuda", generator=g)

    def run(pair_mode):
        C = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        a = _capi.tp_linear_args()
        a.M, a.N, a.K, a.a_k_dup = M, N, 2 * E, E
        a.dtype = a.out_dtype = DT[torch.float16]
        a.lda, a.ldc = E, N
        a.A, a.W, a.C, a.bias = A.data_ptr(), Wp.data_ptr(), C.data_ptr(), bias.data_ptr()
        n0 = lib.tp_debug_counter(_capi.TP_COUNTER_PAIR_LAUNCHES)
        _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, pair_mode)
        try:
            _capi.check(lib.tp_linear(ctypes.byref(a), stream_ptr()), "tp_linear")
        finally:
            _capi.set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)
        torch.cuda.synchronize()
        return C, lib.tp_debug_counter(_capi.TP_COUNTER_PAIR_LAUNCHES) - n0

    c_small, n_small = run(1)
    c_auto, n_auto = run(0)
    assert n_small == 0 and torch.isfinite(c_small.float()).all()
    assert n_auto == (1 if M >= 65536 else 0), (M, n_auto)        # the routing threshold: half a round of the pair kernel's 512 workgroups
    assert torch.equal(c_auto, c_small)
    if M >= 256:                                                   # (TP_TUNE_PAIR_GEMM = 2: wherever the pair kernel supports the launch)
        c_pair, n_pair = run(2)
        if n_pair:
            assert torch.equal(c_pair, c_small), "pair kernel vs 128-tile kernel"
    # fp32 reference: u . (hi_t + lo_t) per 64-wide K-tile
    Wsum = (Wp.float().view(N, E // 64, 2, 64).sum(dim=2)).reshape(N, E)
    ref = A.float() @ Wsum.t() + bias
    err = float((c_small.float() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-3, err
