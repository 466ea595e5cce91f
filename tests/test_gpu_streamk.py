"""Stream-K decomposition of the persistent 256-tile GEMM (tp_gemm8.hip SK; OPT-IN, TP_TUNE_STREAM_K = 2 — measured slower
than the default tail policy, profiles/r03_stream_k_ab.txt): a launch whose tile count is not a multiple of the CU count
shares its K-tiles evenly; a tile cut in two hands one fp32 partial over between neighbouring workgroups.  Checked: against a torch fp32 reference of the same op and against the unsplit kernel (same
value up to the summation order of the fp32 accumulation), determinism under uneven load (another stream holding CUs,
repeated launches: every word identical — the hand-over must never read a stale slab), the whole path at the 8-GPU
shard's batch against the oracle, and the shapes that must NOT take the route."""

import pytest
import torch

from oracle import tokenpacker_oracle as orc
from tests import gpu_util as gu
from tokenpacker_amd import TokenPacker, _capi, synth

pytestmark = pytest.mark.gpu


def _skws():
    lib = _capi.load_library()
    return torch.empty(lib.tp_linear_sk_workspace_bytes(), dtype=torch.uint8, device="cuda")


def _run(A, W, bias, flags, out_dtype, sk, ws, **kw):
    _capi.set_tuning(_capi.TP_TUNE_STREAM_K, 2 if sk else 1)
    try:
        return gu.linear(A, W, bias=bias, flags=flags, out_dtype=out_dtype, sk_workspace=ws if sk else None, **kw)
    finally:
        _capi.set_tuning(_capi.TP_TUNE_STREAM_K, 0)


@pytest.mark.parametrize("M,N,K,dtype,out_dtype,gelu", [
    (18432, 2048, 4096, torch.bfloat16, torch.float16, True),     # the first K/V layer of a 32-image shard: 2.25 tiles per CU
    (4608, 4096, 4096, torch.float16, torch.bfloat16, False),     # mlp[2] of that shard: 1.125 tiles per CU
    (4608, 4096, 4096, torch.float16, torch.float32, False),
    (9216, 4096, 1024, torch.float16, torch.float16, True),       # K = 1024: 16 K-tiles per tile, 36 per workgroup
    (20000, 2048, 4096, torch.bfloat16, torch.float16, True),     # rows not a multiple of the tile
    (10240, 2048, 2048, torch.bfloat16, torch.bfloat16, False),   # 320 tiles, 40 K-tiles per workgroup: cuts snap to boundaries
])
def test_stream_k_matches_reference_and_unsplit_kernel(M, N, K, dtype, out_dtype, gelu):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(dtype)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(dtype)
    bias = torch.randn(N, generator=g, device="cuda") * 0.1
    flags = _capi.TP_LINEAR_GELU if gelu else 0
    ws = _skws()
    y_sk = _run(A, W, bias, flags, out_dtype, True, ws)
    y_plain = _run(A, W, bias, flags, out_dtype, False, ws)
    ref = A.float() @ W.float().t() + bias
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    tol = {torch.float32: 2e-5, torch.float16: 1.5e-3, torch.bfloat16: 1e-2}[out_dtype]
    gu.assert_close(y_sk, ref, f"stream-K {M}x{N}x{K}", tol)
    gu.assert_close(y_plain, ref, f"unsplit {M}x{N}x{K}", tol)
    # same value up to the order of the fp32 accumulation: a few rows round differently, none by more than an output ulp
    d = (y_sk.float() - y_plain.float()).abs().max().item() / ref.abs().max().item()
    frac = (y_sk != y_plain).float().mean().item()
    print(f"\n[stream-K] {M}x{N}x{K} {out_dtype}: max diff vs unsplit {d:.2e} of max|ref|, {100 * frac:.3f} % of elements differ")
    assert d <= {torch.float32: 2e-6, torch.float16: 1.1e-3, torch.bfloat16: 8.5e-3}[out_dtype]
    # (fp32 output shows every re-ordered sum; a 16-bit output only where the rounding flips)
    assert 0 < frac < (0.9 if out_dtype == torch.float32 else 0.2), "stream-K must have run, and only the cut tiles may differ"


def test_stream_k_is_deterministic_under_uneven_load():
    """Repeated launches while another stream holds CUs for random spans: a consumer that read its neighbour's slab before it
    was complete (or a stale line of the previous launch's slab) would change bits.  60 launches, two shapes alternating over
    the SAME slabs, every word compared."""
    lib = _capi.load_library()
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(18432, 2048, 4096), (4608, 4096, 4096)]
    ops = []
    for M, N, K in shapes:
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
        ops.append((A, W, torch.randn(N, generator=g, device="cuda") * 0.1))
    ws = _skws()
    first = [_run(A, W, b, _capi.TP_LINEAR_GELU, torch.float16, True, ws) for A, W, b in ops]
    side = torch.cuda.Stream()
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    import random
    rng = random.Random(3)
    for rep in range(30):
        for k, (A, W, b) in enumerate(ops):
            if rng.random() < 0.7:
                with torch.cuda.stream(side):
                    _capi.check(lib.tp_test_occupy_cus(rng.choice((8, 32, 96)), rng.choice((20, 80, 300)), sink.data_ptr(),
                                                       side.cuda_stream), "occupy")
            y = _run(A, W, b, _capi.TP_LINEAR_GELU, torch.float16, True, ws, sync=False)
            torch.cuda.synchronize()
            assert torch.equal(y, first[k]), f"launch {rep} of shape {shapes[k]} differs: {(y.float() - first[k].float()).abs().max().item()}"


@pytest.mark.parametrize("M,N,K", [(8448, 2048, 4096),      # 264 tiles on 256 CUs: a range would be shorter than a tile + margins
                                   (4608, 4096, 1024),      # 1.125 tiles per CU at K = 1024: same
                                   (147456, 2048, 4096),    # B = 256: 18 tiles per CU exactly — nothing to gain
                                   (2304, 2048, 4096)])     # fewer tiles than CUs
def test_shapes_that_do_not_take_the_route_are_untouched(M, N, K):
    """Off (the default) nothing is decomposed, whatever scratch the caller passes; forced on (TP_TUNE_STREAM_K = 2), an
    ineligible launch runs the unsplit kernels: bit-identical to stream-K off."""
    g = torch.Generator(device="cuda").manual_seed(M)
    A = (torch.randn(min(M, 20000), K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    if M > A.shape[0]:
        A = A.repeat((M + A.shape[0] - 1) // A.shape[0], 1)[:M].contiguous()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    ws = _skws()
    _capi.set_tuning(_capi.TP_TUNE_STREAM_K, 0)                 # the default
    y_auto = gu.linear(A, W, out_dtype=torch.float16, sk_workspace=ws)
    y_off = _run(A, W, None, 0, torch.float16, False, ws)
    assert torch.equal(y_auto, y_off)
    if M != 147456:                                             # (forced on, an ELIGIBLE exact multiple would be decomposed)
        assert torch.equal(_run(A, W, None, 0, torch.float16, True, ws), y_off)


@pytest.mark.parametrize("s,B", [(2, 32), (2, 36), (3, 32), (2, 100)])
def test_whole_path_with_stream_k_against_the_oracle(s, B):
    """The 8-GPU shard (B = 32), the HD shard (36 crops) and a mid-size batch, D = 4096, stream-K on for every eligible launch:
    parity against the fp64 oracle at the shipped gate, run-to-run determinism, and what the default (off) gives."""
    dtype, D = torch.bfloat16, 4096
    params = synth.make_params(400 + s, D)
    x, xm = synth.make_inputs(401 + B, B, dtype)
    m = TokenPacker(hidden_size=D, scale_factor=s)
    m.load_state_dict(params)
    m = m.to(device="cuda", dtype=dtype).eval().requires_grad_(False)
    m.output_fp32 = True
    xg, xmg = x.cuda(), xm.cuda()
    with torch.no_grad():
        y_off = m((xg, xmg))
        _capi.set_tuning(_capi.TP_TUNE_STREAM_K, 2)
        try:
            y = m((xg, xmg))
            y_again = m((xg, xmg))
        finally:
            _capi.set_tuning(_capi.TP_TUNE_STREAM_K, 0)
    torch.cuda.synchronize()
    assert torch.equal(y, y_again)
    p_lp = {k: v.to(dtype) for k, v in params.items()}
    nb = min(B, 6)                                              # the oracle in fp64 on a few images (first and last ones)
    idx = list(range(nb // 2)) + list(range(B - (nb - nb // 2), B))
    y_exact = orc.forward(p_lp, x[idx], xm[idx], scale_factor=s, compute_dtype=torch.float64, io_dtype=dtype)
    e, e_off = orc.rel_err(y[idx], y_exact), orc.rel_err(y_off[idx], y_exact)
    changed = (y != y_off).float().mean().item()
    print(f"\n[stream-K] whole path s={s} B={B}: rel_err {e:.3e} (off: {e_off:.3e}), {100 * changed:.2f} % of output elements differ from stream-K off")
    assert e <= (1.0e-3 if s == 2 else 1.1e-3) and e_off <= (1.0e-3 if s == 2 else 1.1e-3)
    assert changed > 0, "some launch of this batch was expected to be eligible for stream-K"
