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


def _bf16_grads(chain, upstream_scale, seed=77, s=2, D=256, B=3):
    from tokenpacker_amd import TokenPacker
    dtype = torch.bfloat16
    params = synth.make_params(seed, D)
    x, xm = synth.make_inputs(seed + 1, B, dtype)
    w = (torch.randn(B, (24 // s) ** 2, D, generator=torch.Generator().manual_seed(seed + 2)) * upstream_scale).to(dtype)
    m = TokenPacker(hidden_size=D, scale_factor=s)
    m.load_state_dict(params)
    m = m.to(device="cuda", dtype=dtype).train()
    m.tuning = _capi.TuningContext(bwd_chain=chain)
    y = m((x.cuda(), xm.cuda()))
    y.backward(w.cuda())                                     # the upstream gradient IS w (bf16), whatever its magnitude
    torch.cuda.synchronize()
    return {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}, (params, x, xm, w)


def test_fp16_gradient_chain_is_exactly_scale_invariant_and_survives_any_magnitude():
    """TP_TUNE_BWD_CHAIN = 0 (default): a bf16 model's gradients travel in fp16 behind S = 2^k chosen from max |dy| on the device.  An
    upstream gradient multiplied by a power of two therefore gives bit-identical chain values and parameter gradients that are
    EXACTLY that power of two times the original — from 2^-40 (far below fp16's subnormals) to 2^+30 (far above its maximum)."""
    g1, _ = _bf16_grads(0, 1.0)
    for k2 in (-40, -14, 12, 30):
        gk, _ = _bf16_grads(0, 2.0 ** k2)
        for name in g1:
            assert torch.isfinite(gk[name]).all(), (k2, name)
            assert torch.equal(gk[name], (g1[name].double() * 2.0 ** k2).float().to(torch.bfloat16).float()), (k2, name)


def test_both_gradient_chains_against_the_oracle():
    """The fp16 chain (default) and the bf16 chain of rounds 1-5 (TP_TUNE_BWD_CHAIN = 1) against fp64 autograd on the oracle: same
    gradients, the fp16 chain several times closer."""
    from oracle import tokenpacker_oracle as orc2
    worst = {}
    for chain in (0, 1):
        got, (params, x, xm, w) = _bf16_grads(chain, 1.0, seed=91)
        p_lp = {k: v.to(torch.bfloat16) for k, v in params.items()}
        ref_p = {k: v.double().requires_grad_(True) for k, v in p_lp.items()}
        y_ref = orc2.forward(ref_p, x, xm, scale_factor=2, compute_dtype=torch.float64, io_dtype=torch.bfloat16)
        y_ref.backward(w.double())
        want = {k: v.grad for k, v in ref_p.items()}
        rms = {k: float(v.norm()) / v.numel() ** 0.5 for k, v in want.items()}
        errs = []
        for k in want:
            scale = max(rms[k], 0.1 * max(rms[j] for j in want if want[j].shape == want[k].shape))
            errs.append(float((got[k].double() - want[k]).norm()) / want[k].numel() ** 0.5 / scale)
        worst[chain] = max(errs)
        print(f"\n[grad] bf16 model, TP_TUNE_BWD_CHAIN={chain}: worst parameter rel_l2 {worst[chain]:.3e}")
    assert worst[0] <= 1e-2 and worst[1] <= 3e-2 and worst[0] < worst[1]


def test_a_backward_that_leaves_fp16_says_so():
    """The fp16 gradient chain clamps instead of producing inf — and reports it: the backward workspace's status word (bit 0: the
    incoming dy was not finite / a GEMM epilogue clamped, bit 1: LayerNorm backward, bit 2: attention backward), which the module
    reads without synchronising and warns about once.  A clean backward leaves it 0."""
    import warnings
    from tokenpacker_amd import TokenPacker
    dtype, s_, D, B = torch.bfloat16, 2, 256, 2
    params = synth.make_params(31, D)
    x, xm = synth.make_inputs(32, B, dtype)
    m = TokenPacker(hidden_size=D, scale_factor=s_)
    m.load_state_dict(params)
    m = m.to(device="cuda", dtype=dtype).train()
    w = torch.randn(B, 144, D, generator=torch.Generator().manual_seed(33)).to(dtype).cuda()
    m((x.cuda(), xm.cuda())).backward(w)
    torch.cuda.synchronize()
    assert m.backward_saturated() == 0
    m.zero_grad(set_to_none=True)
    w_bad = w.clone()
    w_bad[0, 0, 0] = float("inf")                            # no finite scale brings this into fp16's range
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m((x.cuda(), xm.cuda())).backward(w_bad)
        torch.cuda.synchronize()
        assert m.backward_saturated() & 1
        m.zero_grad(set_to_none=True)
        m((x.cuda(), xm.cuda())).backward(w)                 # the next backward finds the finished copy of the word and warns
        torch.cuda.synchronize()
    assert any("saturated fp16" in str(r.message) for r in rec), [str(r.message) for r in rec]
    assert m.backward_saturated() == 0                       # (the word belongs to ONE backward: this clean one cleared it)


@pytest.mark.parametrize("reserve", [3, 28, 31, 40])
def test_a_forward_on_any_number_of_reserved_cus_is_the_same_function(reserve):
    """TP_TUNE_RESERVE_CUS leaves r CUs per XCD to other streams; since round 6 the persistent kernels take any r (probes run them on 4 CUs per
    XCD; past CUs/8 - 1 one workgroup per XCD is left).  The tile queues make the result independent of how many workgroups draw from them."""
    D, s, B, dtype = 4096, 2, 40, torch.bfloat16
    params = synth.make_params(621, D)
    x, xm = synth.make_inputs(622, B, dtype)
    xg, xmg = x.cuda(), xm.cuda()
    m = _module(params, s, D, dtype)
    with torch.no_grad():
        ref = m((xg, xmg))
        try:
            _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, reserve)
            out = m((xg, xmg))
            torch.cuda.synchronize()
        finally:
            _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, 0)
    assert torch.equal(out, ref)
