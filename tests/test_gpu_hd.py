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
