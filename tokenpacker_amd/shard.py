"""Batch sharding of the projector across the GPUs of one MI355X node.

The path is embarrassingly parallel over images / HD crops: no op mixes batch elements, weights
(73 MB) are replicated.  One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over
xGMI on ROCm; ``gloo`` in the CPU tests).  The ONLY data-path collective is one all-gather of the
projected tokens ``[b_r, M, D]`` so that every rank holds ``[B, M, D]`` "before the LLM"
(BASELINE.json north_star).  The reference has no counterpart (it never calls
torch.distributed); this is new design.

xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), so the gather is issued as a
single large ``all_gather_into_tensor`` (one RCCL call over all links) instead of per-rank
broadcasts; ``overlap_chunks > 1`` splits the local batch so the gather of chunk i overlaps the
kernels of chunk i+1 (RCCL runs on its own stream).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of ``total`` items for ``rank``: the first
    ``total % world_size`` ranks get one extra item (HD crop lists are rarely divisible)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(total: int, world_size: int) -> List[int]:
    return [shard_bounds(total, world_size, r)[1] - shard_bounds(total, world_size, r)[0]
            for r in range(world_size)]


def local_shard(t: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """This rank's slice of a replicated batch tensor (dim 0), as a view (no copy)."""
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(t.shape[0], ws, rk)
    return t[lo:hi]


class GatheredTokens:
    """Result of gathering RAGGED shards (HD crop lists are rarely divisible by the world size) with ONE
    ``all_gather_into_tensor``: ``buf [world * b_max, M, D]`` in which rank r's ``sizes[r]`` items start at row
    block ``r * b_max`` (the last block row of a short rank is padding nobody reads).  Consumers address it IN
    PLACE: ``row_of(i)`` / ``crop_map()`` translate a global item index into its row block (this is what
    ``hd.assemble_hd_tokens(..., crop_map=...)`` reads through), ``g[i]`` is item i, ``compact()`` makes the dense
    ``[total, M, D]`` tensor with one copy when a caller really needs it contiguous."""

    def __init__(self, buf: torch.Tensor, sizes: Sequence[int], b_max: int):
        self.buf, self.sizes, self.b_max = buf, list(sizes), int(b_max)
        self.total = sum(self.sizes)
        self._map: Optional[torch.Tensor] = None

    def row_of(self, i: int) -> int:
        for r, n in enumerate(self.sizes):
            if i < n:
                return r * self.b_max + i
            i -= n
        raise IndexError(i)

    def crop_map(self) -> torch.Tensor:
        """int32 ``[total]`` on the buffer's device: global item index -> row block of ``buf``."""
        if self._map is None:
            idx = [r * self.b_max + j for r, n in enumerate(self.sizes) for j in range(n)]
            self._map = torch.tensor(idx, dtype=torch.int32, device=self.buf.device)
        return self._map

    def __len__(self) -> int:
        return self.total

    def __getitem__(self, i: int) -> torch.Tensor:
        return self.buf[self.row_of(int(i))]

    def compact(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        pieces = [self.buf[r * self.b_max: r * self.b_max + n] for r, n in enumerate(self.sizes)]
        if out is None:
            return torch.cat(pieces, dim=0)
        torch.cat(pieces, dim=0, out=out)
        return out


def ragged_local_buffer(total: int, like: torch.Tensor, tail: Sequence[int], group: Optional[dist.ProcessGroup] = None):
    """``(slot [b_max, *tail], b_r)``: the buffer a ragged shard's projector output is written INTO (its first ``b_r``
    rows), so that the gather needs no pad copy.  ``like`` gives dtype / device."""
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(total, ws)
    return torch.empty((max(sizes),) + tuple(tail), dtype=like.dtype, device=like.device), sizes[rk]


def all_gather_tokens(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None,
                      out: Optional[torch.Tensor] = None, async_op: bool = False, dense: bool = True,
                      slot: Optional[torch.Tensor] = None):
    """Gather ``local [b_r, M, D]`` from every rank (rank order = batch order) with ONE collective.

    Equal shards: one ``all_gather_into_tensor`` straight into ``out [total, M, D]``.
    Ragged shards: one ``all_gather_into_tensor`` of ``b_max``-row slots into a ``[world * b_max, M, D]`` buffer,
    returned as :class:`GatheredTokens` (``dense=False``: addressed in place, no further copy) or compacted into a
    dense tensor (``dense=True``, one copy — needs the gather finished, so it is incompatible with ``async_op``).
    ``slot``: the ``[b_max, M, D]`` buffer ``local`` already lives in (:func:`ragged_local_buffer`) — skips the pad copy.
    Returns the result, and the work handle when ``async_op`` (a real handle in every case: ``work.wait()`` works)."""
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(total, ws)
    b_max = max(sizes)
    tail = tuple(local.shape[1:])
    if local.shape[0] != sizes[rk]:
        raise ValueError(f"local batch {local.shape[0]} != expected shard {sizes[rk]}")
    if min(sizes) == b_max:
        local = local.contiguous()
        if out is None:
            out = torch.empty((total,) + tail, dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op and dense:
        raise ValueError("ragged shards: async_op needs dense=False (compaction must follow the finished gather)")
    if slot is None or slot.shape != (b_max,) + tail or slot.dtype != local.dtype or not slot.is_contiguous() \
            or local.data_ptr() != slot.data_ptr():
        slot = torch.empty((b_max,) + tail, dtype=local.dtype, device=local.device)
        slot[: local.shape[0]] = local                 # (padding rows stay uninitialised: nobody reads them)
    buf = torch.empty((ws * b_max,) + tail, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(buf, slot, group=group, async_op=async_op)
    gathered = GatheredTokens(buf, sizes, b_max)
    if dense:
        return gathered.compact(out)
    return (gathered, work) if async_op else gathered


class TokenGatherPipeline:
    """Software-pipelined all-gather of projected tokens for a STREAM of batches (serving / eval: batch i+1 is
    being projected while batch i's tokens travel).  Each ``submit(local)`` launches ONE asynchronous
    ``all_gather_into_tensor`` into one of ``depth`` rotating ``[total, M, D]`` buffers and returns the slot;
    ``result(slot)`` (or ``drain()``) makes the current stream wait for that gather.  The collective runs on the
    communicator's own stream, so with ``depth >= 2`` it overlaps the next forward's kernels: on one MI355X node
    the gather is per-link bound (every rank receives (P-1)/P of the output over its xGMI links, ~4 ms for
    256 images/GPU) — about the time of the projection itself, i.e. it hides almost entirely.
    Equal shards only (ragged crop lists: ``all_gather_tokens``)."""

    def __init__(self, total: int, group: Optional[dist.ProcessGroup] = None, depth: int = 2):
        ws = dist.get_world_size(group)
        if total % ws != 0:
            raise ValueError(f"TokenGatherPipeline needs equal shards (total {total}, world size {ws})")
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.total, self.group, self.depth = total, group, depth
        self._bufs: List[Optional[torch.Tensor]] = [None] * depth
        self._work = [None] * depth
        self._keep = [None] * depth          # the local shard of an in-flight gather must stay alive
        self._next = 0

    def submit(self, local: torch.Tensor) -> int:
        slot = self._next
        self._next = (slot + 1) % self.depth
        self.result(slot)                    # the buffer's previous gather must be complete before it is re-targeted
        local = local.contiguous()
        shape = (self.total,) + tuple(local.shape[1:])
        buf = self._bufs[slot]
        if buf is None or buf.shape != shape or buf.dtype != local.dtype or buf.device != local.device:
            buf = torch.empty(shape, dtype=local.dtype, device=local.device)
            self._bufs[slot] = buf
        self._keep[slot] = local
        self._work[slot] = dist.all_gather_into_tensor(buf, local, group=self.group, async_op=True)
        return slot

    def result(self, slot: int) -> Optional[torch.Tensor]:
        w = self._work[slot]
        if w is not None:
            w.wait()
            self._work[slot] = None
            self._keep[slot] = None
        return self._bufs[slot]

    def drain(self) -> None:
        for slot in range(self.depth):
            self.result(slot)


def project_sharded(project: Callable[[Tuple[torch.Tensor, torch.Tensor]], torch.Tensor],
                    x_local: torch.Tensor, xm_local: torch.Tensor, total: int,
                    group: Optional[dist.ProcessGroup] = None, gather: bool = True,
                    overlap_chunks: int = 1, dense: bool = True):
    """Run ``project((x, x_multi))`` on this rank's shard and (optionally) all-gather.

    ``project`` is any callable with the projector's forward contract (the HIP ``TokenPacker``
    in production; the CPU tests inject a stand-in).  With ``overlap_chunks > 1`` and equal
    shards, the local shard is processed in chunks and each chunk's gather is launched
    asynchronously so it overlaps the next chunk's kernels.  Ragged shards (``total % world != 0``): a projector
    that can write into a caller's buffer (``project.supports_out``: the HIP module) puts its result straight into
    its gather slot — no pad copy; ``dense=False`` then returns :class:`GatheredTokens` (no compaction copy either)."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    if not gather or ws == 1:
        return project((x_local, xm_local))
    sizes = shard_sizes(total, ws)
    b = x_local.shape[0]
    equal = min(sizes) == max(sizes)
    if not equal:
        if getattr(project, "supports_out", False):
            probe = project.out_like(x_local)                              # (dtype, device, tail) of the result
            slot, b_r = ragged_local_buffer(total, probe[0], probe[1], group)
            y = project((x_local, xm_local), _out=slot[:b_r]) if b_r else slot[:0]
            return all_gather_tokens(y, total, group, dense=dense, slot=slot)
        return all_gather_tokens(project((x_local, xm_local)), total, group, dense=dense)
    if overlap_chunks <= 1 or b % overlap_chunks != 0:
        return all_gather_tokens(project((x_local, xm_local)), total, group)
    cb = b // overlap_chunks
    outs, works = [], []
    for c in range(overlap_chunks):
        y = project((x_local[c * cb:(c + 1) * cb], xm_local[c * cb:(c + 1) * cb]))
        o, w = all_gather_tokens(y, cb * ws, group, async_op=True)
        outs.append(o)
        works.append(w)
    for w in works:
        w.wait()
    # outs[c] is [ws*cb, M, D] in rank order; interleave back to batch order
    M, D = outs[0].shape[1:]
    stacked = torch.stack([o.view(ws, cb, M, D) for o in outs], dim=1)      # [ws, chunks, cb, M, D]
    return stacked.reshape(total, M, D)
