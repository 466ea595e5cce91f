"""CPU oracle for the TokenPacker-HD token assembly.  TEST INFRASTRUCTURE ONLY (see tokenpacker_oracle.py).

Restates the ``mode == 'slice'`` branch of ``prepare_inputs_labels_for_multimodal``
(reference ``llava/model/llava_arch.py:140-154``) for ONE image: crops in row-major order, the ``','`` embedding
between crops of a row, the ``'\\n'`` embedding after every row, then — more than one crop — the global view
and ``'\\n'``.  Parity PINNED: ``tests/golden/hd_splice.npz`` holds what the reference's unmodified
``prepare_inputs_labels_for_multimodal`` returned for a seeded batch (``oracle/make_hd_golden.py:mint_splice``
imports ``llava/model/llava_arch.py`` as it lies and calls it), and ``tests/test_hd_cpu.py`` checks
:func:`splice_inputs_embeds` — which is built on :func:`assemble_one` — against it bit for bit.  The crop-grid choice
(``Image_Patch.calculate``) is pinned by ``tests/golden/hd_grid.json``, minted from the real reference by
``oracle/make_hd_golden.py``."""
from __future__ import annotations

from typing import List, Sequence

import torch


def assemble_one(image_features: torch.Tensor, first_crop: int, h_block: int, w_block: int,
                 sep_embed: torch.Tensor, ret_embed: torch.Tensor):
    """-> (cur_image_features [rows, D], next crop index)   (llava_arch.py:141-154)."""
    pieces = []
    idx = first_crop
    for h in range(h_block):
        for w in range(w_block):
            pieces.append(image_features[idx])
            idx += 1
            if w < w_block - 1:
                pieces.append(sep_embed.reshape(1, -1))
        pieces.append(ret_embed.reshape(1, -1))
    if h_block * w_block > 1:
        pieces.append(image_features[idx])
        pieces.append(ret_embed.reshape(1, -1))
        idx += 1
    return torch.cat(pieces, dim=0), idx


def assemble_hd_tokens(image_features: torch.Tensor, h_block: Sequence[int], w_block: Sequence[int],
                       sep_embed: torch.Tensor, ret_embed: torch.Tensor) -> List[torch.Tensor]:
    out, idx = [], 0
    for hb, wb in zip(h_block, w_block):
        t, idx = assemble_one(image_features, idx, int(hb), int(wb), sep_embed, ret_embed)
        out.append(t)
    assert idx == image_features.shape[0]
    return out


def splice_inputs_embeds(input_ids: torch.Tensor, embed, image_features: torch.Tensor, h_block: Sequence[int],
                         w_block: Sequence[int], sep_id: int, ret_id: int, image_token_index: int = -200) -> torch.Tensor:
    """``new_input_embeds`` of ``prepare_inputs_labels_for_multimodal`` in ``mode == 'slice'`` (llava_arch.py:123-207,
    default flags): per sample, text embeddings up to each image token, that image's assembled block (every image
    token of sample b uses the grid ``h_block[b] x w_block[b]``), the remaining text; a sample without an image
    token still consumes one crop index (:124-134); samples are zero-padded on the right to the longest (:193-200).
    ``embed``: ids -> embeddings.  Returns ``[B, max_len, D]``."""
    sep_e, ret_e = embed(torch.tensor([sep_id])), embed(torch.tensor([ret_id]))
    outs, idx = [], 0
    for b, ids in enumerate(input_ids):
        if int((ids == image_token_index).sum()) == 0:
            outs.append(embed(ids))
            idx += 1
            continue
        pieces, cur = [], ids
        while True:
            pos = torch.where(cur == image_token_index)[0]
            if pos.numel() == 0:
                break
            blk, idx = assemble_one(image_features, idx, int(h_block[b]), int(w_block[b]), sep_e, ret_e)
            pieces += [embed(cur[:int(pos[0])]), blk]
            cur = cur[int(pos[0]) + 1:]
        if cur.numel() > 0:
            pieces.append(embed(cur))
        outs.append(torch.cat(pieces, dim=0))
    L = max(o.shape[0] for o in outs)
    return torch.stack([torch.cat([o, o.new_zeros(L - o.shape[0], o.shape[1])], dim=0) for o in outs], dim=0)


def slice_image(image: torch.Tensor, h_block: int, w_block: int, block: int = 336) -> torch.Tensor:
    """The reference's 'slice' pre-processing for ONE normalised image ``[1, 3, h, w]`` with a GIVEN grid
    (llava/train/train.py:701-731; the grid comes from Image_Patch.calculate, pinned separately): resize
    (torch's own F.interpolate: the third-party arithmetic of this step), zero-pad, tile, global view."""
    import torch.nn.functional as F
    h, w = image.shape[-2:]
    h_ratio = block * h_block / h
    w_ratio = block * w_block / w
    if h_ratio <= w_ratio:
        w_ = min(block * w_block, round(w * h_ratio))
        h_ = block * h_block
    else:
        w_ = block * w_block
        h_ = min(block * h_block, round(h * w_ratio))
    inter = F.interpolate(image, size=(h_, w_), mode="bilinear")
    canvas = torch.zeros((1, 3, block * h_block, block * w_block), dtype=inter.dtype)
    canvas[:, :, :h_, :w_] = inter
    pieces = [canvas[:, :, block * i:block * (i + 1), block * j:block * (j + 1)] for i in range(h_block) for j in range(w_block)]
    if len(pieces) > 1:
        h_ratio = block / h
        w_ratio = block / w
        if h_ratio <= w_ratio:
            w_ = min(block, round(w * h_ratio))
            h_ = block
        else:
            w_ = block
            h_ = min(block, round(h * w_ratio))
        inter = F.interpolate(canvas, size=(h_, w_), mode="bilinear")       # the CANVAS: train.py:710 re-binds `image`
        view = torch.zeros((1, 3, block, block), dtype=inter.dtype)
        view[:, :, :h_, :w_] = inter
        pieces.append(view)
    return torch.cat(pieces, dim=0)
