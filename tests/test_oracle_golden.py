"""The oracle (oracle/tokenpacker_oracle.py) against the golden vectors minted from the REAL
reference module (oracle/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import tokenpacker_oracle as orc
from tokenpacker_amd import synth

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "s[0-9]_D*.npz")))


def _load(path):
    z = np.load(path)
    s, D, B = int(z["scale_factor"]), int(z["hidden_size"]), int(z["batch"])
    params = synth.make_params(int(z["param_seed"]), D)
    x, xm = synth.make_inputs(int(z["input_seed"]), B)
    assert synth.tensor_digest(*params.values()) == str(z["params_sha256"]), "param RNG drift"
    assert synth.tensor_digest(x, xm) == str(z["inputs_sha256"]), "input RNG drift"
    return z, s, D, B, params, x, xm


def test_golden_files_present():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_output(path):
    z, s, D, B, params, x, xm = _load(path)
    y, inter = orc.forward(params, x, xm, scale_factor=s, compute_dtype=torch.float64,
                           return_intermediates=True)
    ostride, istride = int(z["out_row_stride"]), int(z["inter_row_stride"])
    # the reference ran in fp32 on CPU; fp64 oracle agrees to fp32 round-off
    assert orc.rel_err(y[:, ::ostride], torch.from_numpy(z["y"])) < 2e-5
    for key in ("q1", "k1", "v1", "o"):
        got = inter[key][:, ::istride]
        assert orc.rel_err(got, torch.from_numpy(z[key])) < 2e-5, key


@pytest.mark.parametrize("path", [p for p in GOLDEN if "D256" in p],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_fp32_matches_reference_output(path):
    """fp32 oracle (the arithmetic the cpu_baseline leg times) vs the fp32 reference."""
    z, s, D, B, params, x, xm = _load(path)
    y = orc.forward(params, x, xm, scale_factor=s, compute_dtype=torch.float32)
    assert orc.rel_err(y, torch.from_numpy(z["y"])) < 2e-5


@pytest.mark.parametrize("path", [p for p in GOLDEN if "D256" in p],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_reference_low_precision_error_levels(path):
    """Documents the reference's OWN bf16/fp16 deviation from exact math on the same rounded
    weights/inputs (~5e-3 bf16, ~6e-4 fp16) — the yard-stick for the HIP gates."""
    z, s, D, B, params, x, xm = _load(path)
    for tag, dt, lo, hi in (("bf16", torch.bfloat16, 5e-4, 3e-2), ("fp16", torch.float16, 5e-5, 4e-3)):
        p_lp = {k: v.to(dt) for k, v in params.items()}
        y_exact = orc.forward(p_lp, x.to(dt), xm.to(dt), scale_factor=s,
                              compute_dtype=torch.float64, io_dtype=dt)
        e = orc.rel_err(torch.from_numpy(z[f"y_ref_{tag}"]), y_exact)
        assert lo < e < hi, (tag, e)


@pytest.mark.parametrize("s", [1, 2, 3, 4, 6])
def test_bilinear_matches_torch_interpolate(s):
    g = torch.Generator().manual_seed(5)
    grid = torch.randn(2, 24, 24, 16, generator=g, dtype=torch.float64)
    G = 24 // s
    want = torch.nn.functional.interpolate(grid.permute(0, 3, 1, 2), size=(G, G), mode="bilinear")
    got = orc.bilinear_downsample(grid, G)
    assert torch.allclose(got, want.permute(0, 2, 3, 1), atol=1e-12)
    if s <= 4:
        assert torch.allclose(orc.point_queries_closed_form(grid, s), got, atol=1e-12)


def test_region_gather_is_divide_feature():
    """Index form vs the reshape/permute chain of the reference (builder.py:96-105), restated
    here on a small tensor with distinguishable entries."""
    B, g, s, c = 3, 6, 2, 4
    t = torch.arange(B * g * g * c, dtype=torch.float64).reshape(B, g * g, c)
    got = orc.region_gather(t, g, s)                      # [B, G, G, s*s, c]
    G = g // s
    for n in range(B):
        for i in range(G):
            for j in range(G):
                for a in range(s):
                    for b in range(s):
                        tok = (i * s + a) * g + (j * s + b)
                        assert torch.equal(got[n, i, j, a * s + b], t[n, tok])


def test_bad_scale_factor_raises():
    params = synth.make_params(1, 256)
    x, xm = synth.make_inputs(1, 1)
    with pytest.raises(ValueError):
        orc.forward(params, x, xm, scale_factor=5)


def test_oracle_attn_mask_matches_the_reference_module():
    """attn_mask semantics (2-D additive float, 3-D boolean with the reference's region*B+image batch index) pinned on
    outputs of the REAL reference module called with a mask (oracle/make_golden.py ``mask``)."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_s2_D256_B2.npz"))
    s, D, B = int(z["scale_factor"]), int(z["hidden_size"]), int(z["batch"])
    params = synth.make_params(int(z["param_seed"]), D)
    x, xm = synth.make_inputs(int(z["input_seed"]), B)
    m2, m3 = synth.make_attn_masks(int(z["input_seed"]) + 1, B, s)
    assert synth.tensor_digest(x, xm, m2, m3.float()) == str(z["inputs_sha256"])
    for key, mask in (("y_none", None), ("y_2d_float", m2), ("y_3d_bool", m3)):
        y = orc.forward(params, x, xm, scale_factor=s, compute_dtype=torch.float64, attn_mask=mask)
        assert orc.rel_err(y, torch.from_numpy(z[key])) < 2e-5, key
    assert float(np.abs(z["y_3d_bool"] - z["y_none"]).max()) > 0.05        # the masks really change the result


def test_region_major_row_order_is_the_oracles_region_gather():
    """The first K/V layer of the HIP path reads the tower's rows in REGION-MAJOR order (tp_gemm_common.h
    region_major_to_raster, restated here line by line): row r of every K/V-side tensor is then token
    region_gather(...)[b, i, j, kk] with r = b*N + (i*G + j)*s*s + kk — i.e. exactly ``divide_feature``'s grouping
    (builder.py:96-105) flattened, and a bijection of the image's rows."""
    import torch
    from oracle import tokenpacker_oracle as orc

    def region_major_to_raster(row, g, s):
        N, S2, G = g * g, s * s, g // s
        b, t = divmod(row, N)
        q, kk = divmod(t, S2)
        a, c = divmod(kk, s)
        qi, qj = divmod(q, G)
        return b * N + (qi * s + a) * g + qj * s + c

    for g, s, B in ((24, 2, 3), (16, 2, 2), (12, 2, 1), (24, 3, 2), (24, 4, 1), (8, 2, 2)):
        N = g * g
        raster_ids = torch.arange(B * N, dtype=torch.float64).reshape(B, N, 1)
        want = orc.region_gather(raster_ids, g, s).reshape(-1).to(torch.int64).tolist()       # [B, G, G, s*s] flattened
        got = [region_major_to_raster(r, g, s) for r in range(B * N)]
        assert got == want, (g, s)
        assert sorted(got) == list(range(B * N))
