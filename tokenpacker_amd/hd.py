"""TokenPacker-HD helpers on either side of the projector.

* :func:`select_grid` — the crop-grid choice for one image (host-side scalar math; reference
  ``llava/patch_divide.py:71-104`` ``Image_Patch.calculate``).  Pure CPU, like the reference.
* :func:`hd_token_rows`, :func:`assemble_hd_tokens` — what ``prepare_inputs_labels_for_multimodal`` does
  with the projected crops in ``mode == 'slice'`` (reference ``llava/model/llava_arch.py:140-154``): the crops
  of one image row are joined with the ``','`` token embedding, every row and the trailing global view end
  with the ``'\\n'`` embedding.  The reference builds this with a Python loop and one ``torch.cat`` per image;
  here ONE kernel launch (``tp_hd_assemble``) writes all images of the batch straight into one
  ``[rows, D]`` buffer, of which per-image views are returned.  Device tensors only, no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _capi

# Candidate grids (h_block, w_block) in the reference's order — the order decides argmax ties, e.g. between
# (a, b) and (b, a) on square images (patch_divide.py:4-54).  Rule behind the data: every factor pair with
# product <= 9, and for larger products only pairs with both factors >= 2.
_GRIDS_9 = ("1x1 1x2 2x1 1x3 3x1 2x2 1x4 4x1 1x5 5x1 1x6 6x1 2x3 3x2 1x7 7x1 4x2 2x4 1x8 8x1 3x3 1x9 9x1")
_GRIDS_16 = _GRIDS_9 + " 2x5 5x2 2x6 6x2 3x4 4x3 2x7 7x2 3x5 5x3 2x8 8x2 4x4"
_GRIDS_25 = _GRIDS_16 + (" 3x6 6x3 2x9 9x2 4x5 5x4 2x10 10x2 3x7 7x3 11x2 2x11 4x6 6x4 12x2 2x12 3x8 8x3 4x6 6x4 5x5")
_GRID_TABLES = {n: tuple(tuple(int(v) for v in g.split("x")) for g in s.split())
                for n, s in ((9, _GRIDS_9), (16, _GRIDS_16), (25, _GRIDS_25))}


def select_grid(h: int, w: int, patch_num: int = 9, image_size: int = 336) -> Tuple[int, int]:
    """(h_block, w_block) for an ``h x w`` image: the candidate grid maximising
    ``round(h r) round(w r) / area + 0.1 IoU(grid box, 1.4 x image box)`` with ``r`` the largest scale that fits
    the image into the grid box — evaluated in float32 tensor arithmetic exactly as the reference does
    (int64 / int64 true division -> float32, ``torch.round`` half-to-even, first maximum wins)."""
    if patch_num not in _GRID_TABLES:
        raise NotImplementedError(f"patch_num {patch_num} (the reference defines 9, 16, 25)")
    grids = _GRID_TABLES[patch_num]
    size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
    box = torch.tensor([[g[0] * size[0], g[1] * size[1]] for g in grids])           # int64 [n, 2] (far corner)
    area = box[:, 0] * box[:, 1]                                                      # int64
    img = torch.tensor([h, w])
    r = (box / img).min(dim=-1)[0]                                                    # float32
    score = torch.round(h * r) * torch.round(w * r) / area
    img14 = img * 1.4                                                                 # float32
    inter = torch.min(box, img14).clamp(min=0).prod(dim=-1)
    union = area + img14.prod() - inter
    score = score + inter / (union + 1e-5) * 0.1
    return grids[int(torch.argmax(score))]


def slice_plan(h: int, w: int, patch_num: int = 9, block: int = 336):
    """Sizes of the reference's HD slicing for an ``h x w`` image (train.py:699-708, 720-727):
    ``(h_block, w_block, h_res, w_res, hg, wg)`` — the grid, the size the image is resized to inside the
    ``h_block x w_block`` canvas (aspect preserved, Python ``round`` = half-to-even), and the size of the global
    view inside its single block (``0, 0`` when there is only one crop)."""
    hb, wb = select_grid(h, w, patch_num, block)

    def fit(bh, bw):
        h_ratio, w_ratio = bh / h, bw / w
        if h_ratio <= w_ratio:
            return bh, min(bw, round(w * h_ratio))
        return min(bh, round(h * w_ratio)), bw
    h_res, w_res = fit(block * hb, block * wb)
    hg, wg = fit(block, block) if hb * wb > 1 else (0, 0)
    return hb, wb, h_res, w_res, hg, wg


def slice_image(image: torch.Tensor, patch_num: int = 9, block: int = 336):
    """``image [3, H, W]`` fp32 on the GPU (normalised pixels) -> ``(crops [n, 3, block, block] fp32, h_block,
    w_block)`` exactly as the reference's data loader builds ``image_tensor`` in ``'slice'`` mode
    (train.py:695-731): resize + zero-pad + tile, plus the global view when there is more than one crop."""
    if not image.is_cuda:
        raise RuntimeError("slice_image runs only on an AMD GPU (HIP kernel); there is no CPU fallback")
    if image.dim() == 4 and image.shape[0] == 1:
        image = image[0]
    if image.dim() != 3 or image.shape[0] != 3 or image.dtype != torch.float32:
        raise ValueError("image must be a float32 [3, H, W] tensor")
    image = image.contiguous()
    H, W = int(image.shape[1]), int(image.shape[2])
    hb, wb, h_res, w_res, hg, wg = slice_plan(H, W, patch_num, block)
    n = hd_crop_count(hb, wb)
    crops = torch.empty(n, 3, block, block, dtype=torch.float32, device=image.device)
    lib = _capi.load_library()
    with torch.cuda.device(image.device):
        _capi.check(lib.tp_hd_slice(image.data_ptr(), H, W, hb, wb, h_res, w_res, hg, wg, crops.data_ptr(), block,
                                    torch.cuda.current_stream(image.device).cuda_stream), "tp_hd_slice")
    return crops, hb, wb


def hd_token_rows(h_block: int, w_block: int, num_queries: int) -> int:
    """Rows of one image's visual-token block: ``h*w`` crops of ``M`` tokens, a separator after every crop
    (',' inside a row, '\\n' at its end), and — when there is more than one crop — the global view + '\\n'."""
    n = h_block * w_block
    return n * (num_queries + 1) + ((num_queries + 1) if n > 1 else 0)


def hd_crop_count(h_block: int, w_block: int) -> int:
    n = h_block * w_block
    return n + (1 if n > 1 else 0)


def assemble_hd_tokens(image_features: torch.Tensor, h_block: Sequence[int], w_block: Sequence[int],
                       sep_embed: torch.Tensor, ret_embed: torch.Tensor, out: Optional[torch.Tensor] = None,
                       out_rows: Optional[Sequence[int]] = None, crop_map: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """``image_features [n_crops, M, D]`` (projector output for all crops of the batch, crops of image 0 first,
    row-major inside an image, global view last) -> list of per-image ``[rows_i, D]`` views of one buffer, equal
    to the reference's ``cur_image_features`` for each image.

    With ``out [rows, D]`` and ``out_rows`` (first row of each image's block, increasing) the blocks are written
    INTO a caller-owned buffer — e.g. the flattened ``inputs_embeds`` with the text embeddings already in place
    (llava_arch.py:172-191) — and everything between the blocks is left untouched: the visual tokens land where
    the LLM reads them, with no per-image ``torch.cat``.

    ``crop_map`` (int32 device tensor, one entry per crop): crop c is row block ``crop_map[c]`` of ``image_features``
    — the b_max-strided buffer of a ragged all-gather (``shard.GatheredTokens.buf`` / ``.crop_map()``) is read in
    place instead of being compacted first.

    INFERENCE ONLY: the blocks are written by a raw-pointer HIP kernel autograd never sees — with a gradient live on
    ``image_features`` (or on the separator embeddings) the call raises instead of silently cutting the graph between
    the projector and the LLM (the reference's slice branch also runs in training; use its torch.cat form there)."""
    if not image_features.is_cuda:
        raise RuntimeError("assemble_hd_tokens runs only on an AMD GPU (HIP kernel); there is no CPU fallback")
    if torch.is_grad_enabled() and (image_features.requires_grad or sep_embed.requires_grad or ret_embed.requires_grad
                                    or (out is not None and out.requires_grad)):
        raise NotImplementedError(
            "assemble_hd_tokens / build_inputs_embeds write the visual tokens with a HIP kernel outside autograd: with a "
            "gradient live on the projector output (or the separator embeddings / the destination buffer) the graph would "
            "be cut silently.  Call it under torch.no_grad() (inference), or assemble with torch ops when training.")
    if image_features.dim() != 3:
        raise ValueError("image_features must be [n_crops, M, D]")
    if len(h_block) != len(w_block):
        raise ValueError("h_block and w_block must have one entry per image")
    n_crops, M, D = image_features.shape
    if crop_map is not None:
        if crop_map.dtype != torch.int32 or crop_map.dim() != 1 or crop_map.device != image_features.device \
                or not crop_map.is_contiguous():
            raise ValueError("crop_map must be a contiguous int32 vector on the features' device")
        n_crops = crop_map.numel()
    dt = {torch.bfloat16: _capi.TP_BF16, torch.float16: _capi.TP_F16}.get(image_features.dtype)
    if dt is None:
        raise TypeError("image_features must be bfloat16 or float16")
    feats = image_features.contiguous()
    sep = sep_embed.reshape(-1).to(device=feats.device, dtype=feats.dtype).contiguous()
    ret = ret_embed.reshape(-1).to(device=feats.device, dtype=feats.dtype).contiguous()
    if sep.numel() != D or ret.numel() != D:
        raise ValueError(f"separator embeddings must have {D} elements")
    if (out is None) != (out_rows is None):
        raise ValueError("pass `out` and `out_rows` together")
    if out_rows is not None and len(out_rows) != len(h_block):
        raise ValueError("out_rows must have one entry per image")
    plan = (_capi.tp_hd_image * len(h_block))()
    first, row, spans = 0, 0, []
    for i, (hb, wb) in enumerate(zip(h_block, w_block)):
        hb, wb = int(hb), int(wb)
        if hb < 1 or wb < 1:
            raise ValueError("h_block / w_block must be >= 1")
        if out_rows is not None:
            if int(out_rows[i]) < row:
                raise ValueError(f"out_rows[{i}] = {int(out_rows[i])} overlaps the previous image's block (ends at {row})")
            row = int(out_rows[i])
        plan[i] = _capi.tp_hd_image(first, hb, wb, 0, row)
        rows = hd_token_rows(hb, wb, M)
        spans.append((row, rows))
        first += hd_crop_count(hb, wb)
        row += rows
    if first != n_crops:
        raise ValueError(f"h_block/w_block describe {first} crops, image_features holds {n_crops}")
    if out is None:
        out = torch.empty(row, D, dtype=feats.dtype, device=feats.device)
    else:
        if out.dim() != 2 or out.shape[1] != D or out.dtype != feats.dtype or out.device != feats.device \
                or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous [rows, {D}] {feats.dtype} tensor on {feats.device}")
        if row > out.shape[0]:
            raise ValueError(f"the last image's block ends at row {row}, out has {out.shape[0]} rows")
    lib = _capi.load_library()
    with torch.cuda.device(feats.device):
        _capi.check(lib.tp_hd_assemble(plan, len(h_block), feats.data_ptr(), n_crops,
                                       crop_map.data_ptr() if crop_map is not None else None, sep.data_ptr(), ret.data_ptr(),
                                       out.data_ptr(), out.shape[0], M, D, dt,
                                       torch.cuda.current_stream(feats.device).cuda_stream),
                    "tp_hd_assemble")
    return [out[a:a + n] for a, n in spans]


def splice_layout(input_ids: torch.Tensor, h_block: Sequence[int], w_block: Sequence[int], num_queries: int,
                  image_token_index: int = -200):
    """Host-side layout of ``new_input_embeds`` in ``mode == 'slice'`` (llava_arch.py:115-207): for ``input_ids [B, L]``
    returns ``(max_len, text_src, text_dst, img_plan, n_crops)`` — flat source positions (``b * L + pos``) of the text
    tokens, their flat destination rows (``b * max_len + row``), one ``(dest_row, first_crop, h_block, w_block)`` per image
    token in (sample, position) order, and the number of crop indices consumed (a sample without an image token still
    consumes one, :124-134).  Pure integer logic, no device work."""
    ids = input_ids.detach().cpu().tolist()
    per_sample, lengths, crop = [], [], 0
    for b, row_ids in enumerate(ids):
        hb, wb = int(h_block[b]), int(w_block[b])
        segs, row, has_img = [], 0, False
        for pos, tok in enumerate(row_ids):
            if tok == image_token_index:
                has_img = True
                segs.append(("img", row, crop, hb, wb))
                row += hd_token_rows(hb, wb, num_queries)
                crop += hd_crop_count(hb, wb)
            else:
                segs.append(("txt", row, pos))
                row += 1
        if not has_img:
            crop += 1                                     # llava_arch.py:124-134: cur_image_idx += 1
        per_sample.append(segs)
        lengths.append(row)
    L_in, L = (len(ids[0]) if ids else 0), (max(lengths) if lengths else 0)
    text_src, text_dst, img_plan = [], [], []
    for b, segs in enumerate(per_sample):
        for seg in segs:
            if seg[0] == "txt":
                text_src.append(b * L_in + seg[2])
                text_dst.append(b * L + seg[1])
            else:
                img_plan.append((b * L + seg[1], seg[2], seg[3], seg[4]))
    return L, text_src, text_dst, img_plan, crop


def build_inputs_embeds(input_ids: torch.Tensor, embed_tokens, image_features: torch.Tensor, h_block: Sequence[int],
                        w_block: Sequence[int], sep_id: int, ret_id: int, image_token_index: int = -200,
                        crop_map: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``new_input_embeds [B, max_len, D]`` of the reference's ``prepare_inputs_labels_for_multimodal`` in
    ``mode == 'slice'`` (``llava_arch.py:115-207``, default flags), with the visual tokens LANDING IN PLACE.

    The reference walks every sample in Python, embeds text pieces one by one, builds each image's block with a
    loop + ``torch.cat`` (:140-154), concatenates per sample (:190-191) and pads (:193-200).  Here the layout is
    computed once on the host from ``input_ids`` (a tiny D2H copy — the reference synchronises per sample too),
    ALL text tokens of the batch are embedded by one ``embed_tokens`` call and scattered into a zero-initialised
    ``[B * max_len, D]`` buffer, and ONE ``tp_hd_assemble`` launch writes every image block into its rows of that
    same buffer (gaps — the text rows — are left untouched).  ``image_features [n_crops, M, D]`` may be the
    b_max-strided buffer of a ragged all-gather, read through ``crop_map`` (``shard.GatheredTokens``).
    Every image token of sample b uses the grid ``h_block[b] x w_block[b]``; a sample without an image token still
    consumes one crop index, as in the reference (:124-134).  Labels / attention masks stay host-side list logic
    (out of scope).  INFERENCE ONLY (see ``assemble_hd_tokens``): raises when a gradient is live on
    ``image_features`` or on the embedding table."""
    if not image_features.is_cuda:
        raise RuntimeError("build_inputs_embeds runs only on an AMD GPU (HIP kernel); there is no CPU fallback")
    ids_cpu = input_ids.detach().cpu()
    if ids_cpu.dim() != 2 or len(h_block) != ids_cpu.shape[0] or len(w_block) != ids_cpu.shape[0]:
        raise ValueError("input_ids must be [B, L] with one h_block / w_block entry per sample")
    M, D = image_features.shape[1], image_features.shape[2]
    device, dtype = image_features.device, image_features.dtype
    L, text_src, text_dst, img_plan, crop = splice_layout(ids_cpu, h_block, w_block, M, image_token_index)
    n_logical = crop_map.numel() if crop_map is not None else image_features.shape[0]
    if crop != n_logical:
        raise ValueError(f"input_ids / h_block / w_block describe {crop} crops, image_features holds {n_logical}")
    out = torch.zeros(ids_cpu.shape[0] * L, D, dtype=dtype, device=device)
    ids_dev = input_ids.to(device).reshape(-1)
    if text_src:
        src = torch.tensor(text_src, dtype=torch.long, device=device)
        dst = torch.tensor(text_dst, dtype=torch.long, device=device)
        out.index_copy_(0, dst, embed_tokens(ids_dev.index_select(0, src)).to(dtype))
    if img_plan:
        seps = embed_tokens(torch.tensor([sep_id, ret_id], dtype=ids_dev.dtype, device=device)).to(dtype)
        # the plan entries are in (sample, position) order = increasing rows and crops: what tp_hd_assemble requires
        hbs, wbs, rows = [p[2] for p in img_plan], [p[3] for p in img_plan], [p[0] for p in img_plan]
        if crop_map is None and all(p[1] == sum(hd_crop_count(a, c) for a, c in zip(hbs[:i], wbs[:i])) for i, p in enumerate(img_plan)):
            assemble_hd_tokens(image_features, hbs, wbs, seps[0], seps[1], out=out, out_rows=rows)
        else:
            # crops skipped by image-less samples (or a ragged gather buffer): address every crop through a map
            first = torch.tensor([p[1] for p in img_plan])
            logical = torch.cat([torch.arange(int(f), int(f) + hd_crop_count(a, c)) for f, a, c in zip(first, hbs, wbs)])
            cmap = logical.to(device=device, dtype=torch.int32)
            if crop_map is not None:
                cmap = crop_map.index_select(0, cmap.long()).contiguous()
            assemble_hd_tokens(image_features, hbs, wbs, seps[0], seps[1], out=out, out_rows=rows, crop_map=cmap)
    return out.view(ids_cpu.shape[0], L, D)
