"""TokenPacker-HD token assembly on a real MI355X through the C ABI (tp_hd_assemble): bit-exact against the
oracle's restatement of llava_arch.py:140-154, including ragged batches and the single-crop case."""
import pytest
import torch

from oracle import hd_oracle
from tokenpacker_amd import hd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("grids,M,D", [([(1, 1)], 144, 4096), ([(3, 3)], 144, 4096), ([(1, 9), (9, 1), (1, 1), (2, 3)], 64, 256),
                                        ([(2, 2)] * 5 + [(1, 1)] * 70 + [(4, 4)], 36, 512)])
def test_assemble_bit_exact(dtype, grids, M, D):
    g = torch.Generator().manual_seed(5)
    n_crops = sum(hd.hd_crop_count(h, w) for h, w in grids)
    feats = torch.randn(n_crops, M, D, generator=g).to(dtype)
    sep = torch.randn(D, generator=g).to(dtype)
    ret = torch.randn(D, generator=g).to(dtype)
    hb, wb = [h for h, _ in grids], [w for _, w in grids]
    want = hd_oracle.assemble_hd_tokens(feats, hb, wb, sep, ret)
    got = hd.assemble_hd_tokens(feats.cuda(), hb, wb, sep.cuda(), ret.cuda())
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a.cpu(), b)


def test_assemble_argument_errors():
    feats = torch.zeros(3, 4, 16, dtype=torch.bfloat16, device="cuda")
    e = torch.zeros(16, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        hd.assemble_hd_tokens(feats, [1, 1], [1, 1], e, e)            # crop count mismatch (2 != 3)
    with pytest.raises(TypeError):
        hd.assemble_hd_tokens(feats.float(), [1, 1, 1], [1, 1, 1], e, e)
    with pytest.raises(ValueError):
        hd.assemble_hd_tokens(feats, [1, 1, 1], [1, 1, 1], e[:8], e)


@pytest.mark.parametrize("h,w", [(336, 336), (500, 700), (1088, 1088), (300, 1400), (901, 413), (224, 224), (37, 2000)])
def test_slice_image_vs_oracle(h, w):
    """HIP slicing kernel (tp_hd_slice) vs the oracle's torch-CPU F.interpolate restatement of train.py:695-731.
    fp32 bilinear taps in the same formula; torch's vectorised CPU kernel contracts a*b + c*d differently, so the
    bar is a few fp32 ulps of the pixel range rather than bit equality."""
    img = torch.randn(3, h, w, generator=torch.Generator().manual_seed(h * 7 + w))
    crops, hb, wb = hd.slice_image(img.cuda())
    torch.cuda.synchronize()
    want = hd_oracle.slice_image(img.unsqueeze(0), hb, wb)
    assert crops.shape == want.shape and crops.dtype == torch.float32
    err = (crops.cpu() - want).abs().max().item()
    assert err <= 4e-6 * max(1.0, want.abs().max().item()), err
    assert torch.equal(crops.cpu() == 0, want == 0)               # identical zero padding


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_inputs_embeds_match_the_reference_splice(dtype):
    """hd.build_inputs_embeds (one embedding call + ONE tp_hd_assemble launch writing into the inputs_embeds buffer)
    against ``new_input_embeds`` of the reference's unmodified prepare_inputs_labels_for_multimodal, mode 'slice'
    (tests/golden/hd_splice.npz, minted by oracle/make_hd_golden.py): bit-exact — every value is a copy."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
    import make_hd_golden as mk
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hd_splice.npz"))
    ids, hb, wb, feats, table = mk.splice_inputs()
    assert mk._digest(ids, feats, table) == str(z["inputs_sha256"])
    table_d = table.to(dtype).cuda()
    got = hd.build_inputs_embeds(ids.cuda(), lambda t: table_d[t], feats.to(dtype).cuda(), hb, wb,
                                 mk.SPLICE["sep_id"], mk.SPLICE["ret_id"])
    torch.cuda.synchronize()
    want = torch.from_numpy(z["new_input_embeds"])
    assert got.shape == want.shape and got.dtype == dtype
    assert torch.equal(got.float().cpu(), want)          # the minted values are representable in bf16 and fp16


def test_assemble_into_a_buffer_with_gaps_and_through_a_crop_map():
    """out_rows with gaps (text rows of inputs_embeds stay untouched — ADVICE r1: gap rows used to be filled from the
    NEXT image's crops) and crop_map (the b_max-strided buffer of a ragged all-gather read in place)."""
    dtype, M, D = torch.bfloat16, 16, 64
    g = torch.Generator().manual_seed(11)
    grids = [(2, 2), (1, 1), (1, 3)]
    n_crops = sum(hd.hd_crop_count(h, w) for h, w in grids)               # 5 + 1 + 4
    feats = torch.randn(n_crops, M, D, generator=g).to(dtype)
    sep, ret = torch.randn(D, generator=g).to(dtype), torch.randn(D, generator=g).to(dtype)
    hb, wb = [h for h, _ in grids], [w for _, w in grids]
    want = hd_oracle.assemble_hd_tokens(feats, hb, wb, sep, ret)
    rows = [hd.hd_token_rows(h, w, M) for h, w in grids]
    starts = [3, 3 + rows[0] + 7, 3 + rows[0] + 7 + rows[1] + 1]         # gaps of 3, 7 and 1 rows, 5 rows of tail
    total = starts[-1] + rows[-1] + 5
    out = torch.full((total, D), 7.0, dtype=dtype, device="cuda")
    views = hd.assemble_hd_tokens(feats.cuda(), hb, wb, sep.cuda(), ret.cuda(), out=out, out_rows=starts)
    torch.cuda.synchronize()
    mask = torch.ones(total, dtype=torch.bool)
    for st, n, w_, v in zip(starts, rows, want, views):
        assert torch.equal(v.cpu(), w_) and v.data_ptr() == out[st].data_ptr()
        mask[st:st + n] = False
    assert bool((out.cpu()[mask] == 7.0).all()), "rows between / around the image blocks must be left untouched"
    # the same crops scattered over a padded buffer (2 "ranks", b_max = 6: rank 0 holds 6 crops, rank 1 holds 4 + 2 pad)
    padded = torch.full((12, M, D), float("nan")).to(dtype)
    cmap = list(range(6)) + [6 + j for j in range(4)]
    padded[cmap] = feats
    got = hd.assemble_hd_tokens(padded.cuda(), hb, wb, sep.cuda(), ret.cuda(),
                                crop_map=torch.tensor(cmap, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert all(torch.equal(a.cpu(), b) for a, b in zip(got, want))
    with pytest.raises(ValueError):
        hd.assemble_hd_tokens(feats.cuda(), hb, wb, sep.cuda(), ret.cuda(), out=out, out_rows=[3, 10, 500])   # overlap
