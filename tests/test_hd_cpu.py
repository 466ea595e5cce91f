"""TokenPacker-HD host logic on CPU: crop-grid selection against goldens minted from the reference's
``Image_Patch.calculate`` (oracle/make_hd_golden.py), row bookkeeping against the oracle's concatenation, and
the C ABI's row count."""
import json
import os

import pytest
import torch

from oracle import hd_oracle
from tokenpacker_amd import _capi, hd

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hd_grid.json")


@pytest.mark.parametrize("patch_num", [9, 16, 25])
def test_select_grid_matches_reference_choices(patch_num):
    z = json.load(open(GOLD))
    assert len(hd._GRID_TABLES[patch_num]) == z["candidates"][str(patch_num)]
    wrong = [(hw, got, want) for hw, want in zip(z["sizes"], z["choices"][str(patch_num)])
             if (got := list(hd.select_grid(hw[0], hw[1], patch_num))) != want]
    assert not wrong, wrong[:5]


def test_select_grid_known_cases_and_errors():
    assert hd.select_grid(1088, 1088, 9) == (3, 3)            # 1088x1088 -> 3x3 grid
    assert hd.select_grid(336, 336, 9) == (1, 1)
    with pytest.raises(NotImplementedError):                   # patch_divide.py:80
        hd.select_grid(100, 100, 10)


@pytest.mark.parametrize("h,w", [(1, 1), (1, 2), (2, 1), (3, 3), (1, 9), (9, 1), (2, 4), (5, 5)])
@pytest.mark.parametrize("M", [36, 64, 144])
def test_row_count_matches_oracle_and_abi(h, w, M):
    D = 16
    n = hd.hd_crop_count(h, w)
    feats = torch.arange(n * M * D, dtype=torch.float32).reshape(n, M, D)
    ref, nxt = hd_oracle.assemble_one(feats, 0, h, w, torch.full((D,), -1.0), torch.full((D,), -2.0))
    assert nxt == n and ref.shape[0] == hd.hd_token_rows(h, w, M)
    assert _capi.load_library().tp_hd_rows(h, w, M) == ref.shape[0]
    # BASELINE config 4: a 3x3 grid + global view at s=2 is 10 crops and 1450 LLM tokens
    if (h, w, M) == (3, 3, 144):
        assert ref.shape[0] == 1450


def test_assemble_rejects_cpu_tensors():
    with pytest.raises(RuntimeError):
        hd.assemble_hd_tokens(torch.zeros(1, 4, 8, dtype=torch.bfloat16), [1], [1], torch.zeros(8), torch.zeros(8))


def test_slice_oracle_matches_reference_dataloader_code():
    """oracle.hd_oracle.slice_image against crops produced by the reference's own 'slice' code
    (train.py:695-731, executed from the source text by oracle/make_hd_golden.py): bit-exact, both sides run
    torch-CPU F.interpolate.  Also pins hd.slice_plan's grid and sizes."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hd_slice.npz"))
    stride = int(z["stride"])
    for k, (h, w) in enumerate(z["sizes"].tolist()):
        img = torch.randn(3, h, w, generator=torch.Generator().manual_seed(1000 + k))
        hb, wb = [int(v) for v in z[f"grid_{k}"]]
        plan = hd.slice_plan(h, w)
        assert plan[:2] == (hb, wb)
        crops = hd_oracle.slice_image(img.unsqueeze(0), hb, wb)
        assert crops.shape[0] == hd.hd_crop_count(hb, wb)
        assert np.array_equal(crops[:, :, ::stride, ::stride].numpy(), z[f"sub_{k}"])
        s = z[f"sum_{k}"]
        assert float(crops.double().sum()) == s[0] and float(crops.double().abs().sum()) == s[1]
        # the resized extent implied by the plan is where the canvas stops being zero-padded
        assert plan[2] <= 336 * hb and plan[3] <= 336 * wb and (plan[2] == 336 * hb or plan[3] == 336 * wb)


def test_splice_oracle_matches_the_unmodified_prepare_inputs_labels_for_multimodal():
    """``hd_oracle.splice_inputs_embeds`` (and through it ``assemble_one``) against the output of the reference's own
    function (llava_arch.py:100-233, mode 'slice'), called unmodified at mint time (oracle/make_hd_golden.py)."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
    import make_hd_golden as mk
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hd_splice.npz"))
    ids, hb, wb, feats, table = mk.splice_inputs()
    assert mk._digest(ids, feats, table) == str(z["inputs_sha256"])
    got = hd_oracle.splice_inputs_embeds(ids, lambda t: table[t], feats, hb, wb, mk.SPLICE["sep_id"], mk.SPLICE["ret_id"])
    want = torch.from_numpy(z["new_input_embeds"])
    assert got.shape == want.shape and torch.equal(got, want)


def test_splice_layout_reproduces_the_reference_embeds_on_cpu():
    """hd.splice_layout (the host-side integer logic of hd.build_inputs_embeds) drives a plain torch-CPU reconstruction —
    text rows scattered by the layout, image blocks by the oracle's assemble_one at the layout's rows — that must equal
    the reference's ``new_input_embeds`` bit for bit.  (The GPU test checks the kernel that fills the same layout.)"""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle"))
    import make_hd_golden as mk
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hd_splice.npz"))
    ids, hb, wb, feats, table = mk.splice_inputs()
    M, D = feats.shape[1], feats.shape[2]
    L, text_src, text_dst, img_plan, n_crops = hd.splice_layout(ids, hb, wb, M)
    want = torch.from_numpy(z["new_input_embeds"])
    assert L == want.shape[1] and n_crops == feats.shape[0]
    out = torch.zeros(ids.shape[0] * L, D)
    out[torch.tensor(text_dst)] = table[ids.reshape(-1)[torch.tensor(text_src)]]
    sep, ret = table[mk.SPLICE["sep_id"]], table[mk.SPLICE["ret_id"]]
    prev_end = -1
    for row, first, h, w in img_plan:
        blk, nxt = hd_oracle.assemble_one(feats, first, h, w, sep, ret)
        assert blk.shape[0] == hd.hd_token_rows(h, w, M) and nxt - first == hd.hd_crop_count(h, w) and row > prev_end
        out[row:row + blk.shape[0]] = blk
        prev_end = row + blk.shape[0] - 1
    assert torch.equal(out.view(ids.shape[0], L, D), want)
    # the image-less sample consumed one crop index: the crops of the last sample start one later
    assert [p[1] for p in img_plan] == [0, 7, 8, 10]
