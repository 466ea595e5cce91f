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


class DirectGather:
    """The all-gather of projected tokens WITHOUT a collective kernel: every rank's shard travels to every peer as one
    ``hipMemcpyAsync`` per peer (SDMA engines over that peer's xGMI link — no compute unit), into rows ``[lo, hi)`` of
    the peer's receive buffer, which the peer exported once through HIP IPC (``tp_gather_*`` in include/tokenpacker.h).

    Why: RCCL's all-gather runs as kernels, and the next forward's persistent GEMMs own every CU (one 512-thread
    workgroup with the whole register file per CU) — a collective that overlaps the next forward either waits for CUs or
    takes them from the GEMMs (16 busy CUs: +19 % per forward, profiles/r02y_hog_bench_B256.json).  Copy engines need none.

    Protocol (one instance per rank, same arguments everywhere; ``depth`` rotating receive buffers ``[total, M, D]``)::

        view = g.begin()                      # rows [lo, hi) of this step's buffer: where the projector writes
        model((x, x_multi), _out=view)        # (or g.submit(local) copies a shard that lives elsewhere)
        t = g.submit()                        # sync kernel (1 wave) + one copy and one 4-byte flag per peer, all enqueued
        ...                                   # next forward(s): the copies ride the SDMA engines meanwhile
        tokens = g.result(t)                  # [total, M, D]; the current stream waits for every peer's flag of that step

    Sequence numbers instead of acknowledgements: peer p's flag of step i lands here behind p's shard of step i.  ``submit``
    of step i first waits (on the current stream) until every peer's flag of step i-1 is here — which also proves that each
    peer has passed ITS submit of step i-1, i.e. has finished with the buffer step ``i - 1 + 1 - depth`` lived in.  Hence
    the CONTRACT: the reads of a step's tokens are enqueued (on the stream ``submit`` is called on) before
    ``submit`` number ``depth - 1`` after it — with the default depth 3, "before the second-next submit".
    Ragged shards are fine (every rank writes its own row range).  ``use_cus=True`` lets the runtime pick blit kernels
    instead of SDMA (A/B)."""

    def __init__(self, total: int, tail: Sequence[int], dtype: torch.dtype, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None, depth: int = 3, use_cus: bool = False, timeout_ms: int = 30000):
        import ctypes
        from . import _capi
        if depth < 2:
            raise ValueError("DirectGather needs depth >= 2 (a buffer is being filled while the previous one is read)")
        if depth > 6:
            # the sequence numbers travel through an 8-slot cell ring (cell seq % 8): a peer stream's 4-byte flag copy of step i may
            # still be pending `depth` steps later; the cell of step i is re-written by step i + 8, so depth 8 is where a pending
            # copy would collide — depth 7 is the last that cannot, and the bound is kept one further below that (<= 6) as margin
            # for the poll of step i - 1 that a rank performs while its own step i + depth - 1 copies are already queued
            raise ValueError("DirectGather: depth <= 6 (the sequence cells are an 8-slot ring; 8 collides, 7 has no margin)")
        self._capi, self._ct = _capi, ctypes
        self.lib = _capi.load_library()
        self.group, self.depth, self.use_cus, self.timeout_ms = group, depth, int(bool(use_cus)), int(timeout_ms)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 64:
            raise ValueError("DirectGather: at most 64 ranks (one wave polls the flags)")
        self.device = torch.device(device)
        self.sizes = shard_sizes(total, self.world)
        self.lo, self.hi = shard_bounds(total, self.world, self.rank)
        self.total, self.tail = total, tuple(tail)
        # Set-up is COLLECTIVE-SAFE: a rank whose local step fails (an IPC export / mapping the driver refuses) still takes part
        # in the exchanges below, and then EVERY rank raises — nobody is left waiting in a barrier for a rank that has gone.
        self._opened = []                                     # mapped allocation bases (tp_gather_close at close())
        self.peers = [p for p in range(self.world) if p != self.rank]
        err, mine = None, None
        try:
            with torch.cuda.device(self.device):
                self.bufs = [torch.empty((total,) + self.tail, dtype=dtype, device=self.device) for _ in range(depth)]
                # flags: FINE-GRAINED memory from the library (polled by a kernel here while peers' copy engines write it);
                # where the runtime refuses it, ordinary memory — correct between processes on one device, documented as such
                fp = ctypes.c_void_p()
                if self.lib.tp_gather_alloc_flags(ctypes.byref(fp), 64 * 4) == _capi.TP_OK:
                    self._flags_ptr, self._flags_owned, self._flags_keep = int(fp.value), True, None
                else:
                    import warnings
                    warnings.warn("DirectGather: the runtime refused fine-grained memory for the sequence flags; falling back to ordinary "
                                  "device memory, which is only guaranteed coherent for flags written by OTHER devices' copy engines at "
                                  "kernel boundaries — fine between processes on one device, on a multi-GPU node expect stale flags "
                                  "(timeouts): use the rccl gather there", RuntimeWarning)
                    self._flags_keep = torch.zeros(64, dtype=torch.int32, device=self.device)
                    self._flags_ptr, self._flags_owned = self._flags_keep.data_ptr(), False
                self.cells = torch.zeros(8, dtype=torch.int32, device=self.device)
                self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
                torch.cuda.synchronize(self.device)           # the zeros are in place before any peer can write a flag
                row_bytes = self.bufs[0][0].numel() * self.bufs[0].element_size() if total else 0
                self._nbytes = (self.hi - self.lo) * row_bytes
                mine = {"bufs": [self._export(b.data_ptr()) for b in self.bufs], "flags": self._export(self._flags_ptr)}
        except Exception as exc:                              # noqa: reported to every rank below
            err = f"rank {self.rank}: {exc!r}"
        everyone = [None] * self.world
        dist.all_gather_object(everyone, {"err": err, "handles": mine}, group=group)
        errs = [e["err"] for e in everyone if e["err"]]
        if errs:
            raise RuntimeError("DirectGather: exporting the receive buffers failed: " + "; ".join(errs))
        try:
            with torch.cuda.device(self.device):
                self._dst = [[0] * len(self.peers) for _ in range(depth)]     # [buffer][peer] -> address of MY rows in the peer's buffer
                self._dst_flag = [0] * len(self.peers)                        # address of flags[self.rank] in the peer's flag array
                for j, p in enumerate(self.peers):
                    bases = {}

                    def mapped(rec, bases=bases):
                        handle, off = rec
                        if handle not in bases:
                            base = ctypes.c_void_p()
                            hb = (ctypes.c_char * _capi.TP_IPC_HANDLE_BYTES).from_buffer_copy(handle)
                            _capi.check(self.lib.tp_gather_open(hb, ctypes.byref(base)), "tp_gather_open")
                            bases[handle] = base.value
                            self._opened.append(base.value)
                        return bases[handle] + off
                    for k in range(depth):
                        self._dst[k][j] = mapped(everyone[p]["handles"]["bufs"][k]) + self.lo * row_bytes
                    self._dst_flag[j] = mapped(everyone[p]["handles"]["flags"]) + 4 * self.rank
                self.streams = [torch.cuda.Stream(device=self.device) for _ in self.peers]
        except Exception as exc:                              # noqa
            err = f"rank {self.rank}: {exc!r}"
        oks = [None] * self.world
        dist.all_gather_object(oks, err, group=group)          # (also the barrier: every rank has mapped every peer before anyone pushes)
        if any(oks):
            self._unmap()
            raise RuntimeError("DirectGather: mapping the peers' buffers failed: " + "; ".join(e for e in oks if e))
        self.flags_fine_grained = bool(self._flags_owned)     # (echoed by bench.py: False = the runtime refused fine-grained memory)
        self._push_done = [[None] * len(self.peers) for _ in range(depth)]
        self._step = 0
        self._waited = 0                                      # highest sequence number a sync on the current stream has covered
        self._closed = False

    def _unmap(self) -> None:
        with torch.cuda.device(self.device):
            for base in self._opened:
                self.lib.tp_gather_close(base)
        self._opened = []

    def _export(self, ptr: int):
        ct, _capi = self._ct, self._capi
        handle = (ct.c_char * _capi.TP_IPC_HANDLE_BYTES)()
        off = ct.c_uint64(0)
        _capi.check(self.lib.tp_gather_export(ptr, handle, ct.byref(off)), "tp_gather_export")
        return bytes(handle), int(off.value)

    def _sync(self, wait_seq: int, cell: Optional[int], publish: int) -> None:
        cur = torch.cuda.current_stream(self.device)
        self._capi.check(self.lib.tp_gather_sync(self._flags_ptr, self.world, self.rank, wait_seq & 0xFFFFFFFF,
                                                 cell, publish & 0xFFFFFFFF, self.status.data_ptr(), self.timeout_ms,
                                                 cur.cuda_stream), "tp_gather_sync")

    def begin(self) -> torch.Tensor:
        """Rows ``[lo, hi)`` of the buffer the NEXT ``submit`` sends: the projector's ``_out``.  (The current stream first
        waits for this rank's own copies out of that buffer, ``depth`` steps ago.)"""
        k = self._step % self.depth
        cur = torch.cuda.current_stream(self.device)
        for ev in self._push_done[k]:
            if ev is not None:
                cur.wait_event(ev)
        return self.bufs[k][self.lo:self.hi]

    def submit(self, local: Optional[torch.Tensor] = None):
        """Send this step's shard to every peer; returns the ticket ``result`` takes.  ``local``: a shard that is not
        already in ``begin()``'s view is copied there first."""
        ct = self._ct
        with torch.cuda.device(self.device):
            view = self.begin()
            if local is not None and (local.data_ptr() != view.data_ptr() or tuple(local.shape) != tuple(view.shape)):
                if tuple(local.shape) != tuple(view.shape) or local.dtype != view.dtype:
                    raise ValueError(f"local shard {tuple(local.shape)} {local.dtype} != expected {tuple(view.shape)} {view.dtype}")
                view.copy_(local)
            k, seq = self._step % self.depth, self._step + 1
            cell = self.cells.data_ptr() + 4 * (seq % 8)
            cur = torch.cuda.current_stream(self.device)
            self._sync(seq - 1, cell, seq)                    # every peer's previous shard is here; publish this step's number
            self._waited = max(self._waited, seq - 1)
            n = len(self.peers)
            if n:
                ev = torch.cuda.Event()
                ev.record(cur)
                for st in self.streams:
                    st.wait_event(ev)
                arr = ct.c_void_p * n
                self._capi.check(self.lib.tp_gather_push(n, arr(*self._dst[k]), view.data_ptr(), self._nbytes, arr(*self._dst_flag),
                                                         cell, arr(*[st.cuda_stream for st in self.streams]), self.use_cus),
                                 "tp_gather_push")
                for j, st in enumerate(self.streams):
                    e = torch.cuda.Event()
                    e.record(st)
                    self._push_done[k][j] = e
            self._step += 1
            return (seq, k)

    def result(self, ticket) -> torch.Tensor:
        """``[total, M, D]`` of the step ``ticket`` names; the current stream waits until every peer's shard has landed."""
        seq, k = ticket
        if seq > self._waited:
            with torch.cuda.device(self.device):
                self._sync(seq, None, 0)
            self._waited = seq
        return self.bufs[k]

    def drain(self) -> None:
        if self._step:
            self.result((self._step, (self._step - 1) % self.depth))

    def check(self) -> None:
        """Raise if a wait ran into its timeout (a peer died or never submitted).  Synchronises."""
        if int(self.status.item()) != 0:
            raise TimeoutError(f"DirectGather rank {self.rank}: a peer's shard did not arrive within {self.timeout_ms} ms")

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        self.drain()
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)                        # nobody is still writing into anybody
        self._unmap()
        if self._flags_owned:
            with torch.cuda.device(self.device):
                self.lib.tp_gather_free_flags(self._flags_ptr)
            self._flags_owned = False
        self.check()


def project_sharded(project: Callable[[Tuple[torch.Tensor, torch.Tensor]], torch.Tensor],
                    x_local: torch.Tensor, xm_local: torch.Tensor, total: int,
                    group: Optional[dist.ProcessGroup] = None, gather: bool = True,
                    overlap_chunks: int = 1, dense: bool = True, force_collective: bool = False):
    """Run ``project((x, x_multi))`` on this rank's shard and (optionally) all-gather.

    ``project`` is any callable with the projector's forward contract (the HIP ``TokenPacker``
    in production; the CPU tests inject a stand-in).  With ``overlap_chunks > 1`` and equal
    shards, the local shard is processed in chunks and each chunk's gather is launched
    asynchronously so it overlaps the next chunk's kernels.  Ragged shards (``total % world != 0``): a projector
    that can write into a caller's buffer (``project.supports_out``: the HIP module) puts its result straight into
    its gather slot — no pad copy; ``dense=False`` then returns :class:`GatheredTokens` (no compaction copy either)."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    # (a one-rank group has nothing to gather; ``force_collective`` sends it through the collective all the same — the one-rank
    # RCCL smoke of the GPU suite and ``bench.py --force-dist``: the library loads, the communicator binds, the copy is exact)
    if not gather or (ws == 1 and not (force_collective and dist.is_initialized())):
        return project((x_local, xm_local))
    sizes = shard_sizes(total, ws)
    b = x_local.shape[0]
    equal = min(sizes) == max(sizes)
    if not equal:
        # (the _out fast path is inference-only: TokenPacker refuses _out while a gradient is live — then the shard goes
        # through the plain forward and the pad copy, as it did before the fast path existed)
        grad_live = torch.is_grad_enabled() and any(p.requires_grad for p in getattr(project, "parameters", lambda: [])())
        if getattr(project, "supports_out", False) and not grad_live:
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
