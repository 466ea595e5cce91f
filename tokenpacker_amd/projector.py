"""Drop-in ``TokenPacker`` projector backed by hand-written HIP kernels for MI355X (gfx950).

Boundary mirrored: reference
``llava/model/multimodal_projector/builder.py:39-145`` —

* same constructor signature and defaults as the reference ``TokenPacker`` (builder.py:40-49);
* same 23 parameters under the same state-dict names (so ``load_state_dict`` of a reference
  ``mm_projector.bin`` works unchanged, llava_arch.py:78-83) — the parameters live in ordinary
  ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.MultiheadAttention`` containers, which this module never
  *calls*;
* ``forward(x, attn_mask=None)`` takes the tuple ``(x, x_multi)`` the CLIP tower returns
  (clip_encoder.py:62) and yields ``[B, (raw_grid//scale_factor)**2, hidden_size]`` in ``x[0]``'s
  dtype on ``x[0]``'s device (builder.py:107-137), so it can sit behind the unmodified
  ``encode_images()`` (llava_arch.py:95-98);
* ``build_vision_projector(config)`` reads ``config.hidden_size`` / ``config.scale_factor``
  exactly like builder.py:144-145.

All arithmetic happens in ``libtokenpacker_hip.so`` through the C ABI of
``include/tokenpacker.h``; torch only owns memory and the stream.  There is no eager / CPU
fallback: CPU tensors, unsupported dtypes or a missing library raise.
"""
from __future__ import annotations

import ctypes
import weakref
from functools import partial
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _capi

_DTYPES = {torch.bfloat16: _capi.TP_BF16, torch.float16: _capi.TP_F16}


class _ProjectFn(torch.autograd.Function):
    """Autograd node of the HIP projector: ``tp_forward_train`` keeps the backward's operands in a per-call
    workspace, ``tp_backward`` returns the gradients of the 23 parameters (the CLIP features come from a frozen
    ``no_grad`` tower — reference clip_encoder.py:46, train.py:950-953 — and get none)."""

    @staticmethod
    def forward(ctx, module, x, n_parts, *rest):
        xm, params = rest[:n_parts], rest[n_parts:]          # x_multi: one tensor, or its four sources
        out, train_ws, desc, packed = module._launch_forward(x, xm[0] if n_parts == 1 else xm, train=True)
        ctx.module, ctx.desc, ctx.n_parts = module, desc, n_parts
        ctx.save_for_backward(train_ws, packed, *xm, *params)
        return out

    @staticmethod
    def backward(ctx, dy):
        train_ws, packed, *rest = ctx.saved_tensors
        xm, params = rest[:ctx.n_parts], rest[ctx.n_parts:]
        grads = ctx.module._launch_backward(ctx.desc, xm[0] if ctx.n_parts == 1 else xm, train_ws, packed, params, dy)
        return (None, None, None) + (None,) * ctx.n_parts + tuple(grads)


def _forget_packed(ptr: int) -> None:
    try:
        _capi.load_library().tp_pack_forget(ptr)
    except Exception:                # noqa: interpreter shutdown
        pass


import contextlib as _contextlib
_NULLCTX = _contextlib.nullcontext()


class TokenPacker(nn.Module):
    """Region-to-point visual projector (see module docstring)."""

    MULTI_LEVEL_DIM = 4096      # builder.py:61,67 hard-code nn.Linear(4096, 1024)
    supports_out = True         # forward(..., _out=buffer): tokenpacker_amd.shard writes ragged shards into their gather slot

    def __init__(self, raw_grid: int = 24, embed_dim: int = 1024, num_heads: int = 1024 // 128,
                 kv_dim: int = 1024, hidden_size: int = 4096, scale_factor: int = 2,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6)):
        super().__init__()
        if raw_grid % scale_factor != 0:
            # same exception type and message as builder.py:51-52
            raise ValueError("scale_factor must be divisible by grid size")
        if embed_dim != 1024 or kv_dim != 1024 or num_heads != 8:
            raise ValueError("the HIP kernels are specialised for embed_dim=kv_dim=1024, num_heads=8 "
                             "(the only configuration the reference instantiates)")
        if hidden_size % 128 != 0:
            raise ValueError("hidden_size must be a multiple of 128")
        self.raw_grid = raw_grid
        self.grid_size = raw_grid // scale_factor
        self.num_queries = self.grid_size ** 2
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.scale_factor = scale_factor
        self.hidden_size = hidden_size

        # ---- parameter containers (names/shapes are the state-dict contract) ----
        self.q_proj_1 = nn.Linear(kv_dim, embed_dim, bias=False)
        self.k_proj_1 = nn.Sequential(nn.Linear(self.MULTI_LEVEL_DIM, 1024), nn.GELU(), nn.Linear(1024, 1024))
        self.v_proj_1 = nn.Sequential(nn.Linear(self.MULTI_LEVEL_DIM, 1024), nn.GELU(), nn.Linear(1024, 1024))
        self.ln_q_1 = norm_layer(embed_dim)
        self.ln_k_1 = norm_layer(embed_dim)
        self.ln_v_1 = norm_layer(embed_dim)
        self.clip_attn = nn.MultiheadAttention(embed_dim, num_heads)
        self.mlp = nn.Sequential(nn.Linear(1024, hidden_size), nn.GELU(), nn.Linear(hidden_size, hidden_size))
        self._reset_parameters()

        #: set True to receive fp32 output straight from the last GEMM's accumulators
        #: (validation mode of SURVEY.md §8c; default = input dtype like the reference)
        self.output_fp32 = False
        #: fp32 callers.  The reference module runs in any dtype; the HIP kernels compute in bf16 / fp16 MFMA.  Under
        #: ``torch.autocast`` an fp32 module (fp32 master weights) is served in the autocast dtype like every
        #: ``nn.Linear`` of the reference would be.  Outside autocast an fp32 module is refused unless this names the
        #: dtype to compute in (torch.float16 / torch.bfloat16): operands are rounded to it, the result comes back
        #: as fp32 from the last GEMM's accumulators (within 1e-3 of the fp32 reference, not bit-comparable to it).
        self.fp32_compute_dtype: Optional[torch.dtype] = None
        #: a ``_capi.TuningContext`` this module's calls read their knobs from (``tp_desc.tuning``): a serving worker that runs
        #: several model instances / threads gives each its own, and nobody's ``set_tuning`` can change another's schedule or
        #: low bits.  None (default): the library's process-wide table.  Not copied by deepcopy / pickle.
        self.tuning = None
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._packed_event = None
        self._packed_stream = None
        self._overflow_checked = False
        self._workspaces: Dict[tuple, torch.Tensor] = {}
        self._last_launch = None             # (desc, workspace) of the last inference forward: saturation_report()
        self._sat_warned, self._sat_pending, self._sat_count = False, None, 0
        self._bwd_sat_warned, self._bwd_sat_pending = False, None      # the backward workspace's status word (fp16 gradient chain)

    # ------------------------------------------------------------------------------------------
    _CACHE_DEFAULTS = {"tuning": None, "_packed": None, "_packed_key": None, "_packed_event": None, "_packed_stream": None,
                       "_overflow_checked": False, "_last_launch": None, "_sat_warned": False, "_sat_pending": None,
                       "_sat_count": 0, "_bwd_sat_warned": False, "_bwd_sat_pending": None}

    def __getstate__(self):
        """``copy.deepcopy`` / ``pickle`` / ``torch.save(module)`` carry the parameters and settings, never the kernel-side
        caches (packed weight image, workspaces, HIP event): a copy re-packs on its first forward."""
        state = self.__dict__.copy()
        state.update(self._CACHE_DEFAULTS)
        state["_workspaces"] = {}
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        for k, v in self._CACHE_DEFAULTS.items():
            self.__dict__.setdefault(k, v)
        self.__dict__.setdefault("_workspaces", {})

    def _reset_parameters(self) -> None:
        """Same initial distribution as the reference (builder.py:87-94): truncated normal
        (std 0.02) for every nn.Linear weight (out_proj included), zero biases, unit LayerNorm;
        ``clip_attn.in_proj_weight`` keeps nn.MultiheadAttention's own xavier init."""
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                nn.init.trunc_normal_(mod.weight, std=0.02)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)
            elif isinstance(mod, nn.LayerNorm):
                nn.init.ones_(mod.weight)
                nn.init.zeros_(mod.bias)

    # ------------------------------------------------------------------------------------------
    _WEIGHT_PATHS = tuple(tuple(name.split(".")) for name in _capi.WEIGHT_FIELDS)

    def _named_weights(self):
        """The 23 parameters in tp_weights order.  Walks the registration dicts directly (``dict(self.named_parameters())`` costs 25 us
        per call — a fifth of a one-image forward, which is bound by the host's enqueue rate: profiles/r06p_timeline_B1.txt);
        always the CURRENT objects, so a replaced submodule / parameter is seen."""
        out = []
        for path in self._WEIGHT_PATHS:
            mod = self
            for key in path[:-1]:
                mod = mod._modules[key]
            out.append(mod._parameters[path[-1]])
        return out

    def _expected_shapes(self):
        E, C, D = self.embed_dim, self.MULTI_LEVEL_DIM, self.hidden_size
        return {"q_proj_1.weight": (E, E),
                "k_proj_1.0.weight": (E, C), "k_proj_1.0.bias": (E,), "k_proj_1.2.weight": (E, E), "k_proj_1.2.bias": (E,),
                "v_proj_1.0.weight": (E, C), "v_proj_1.0.bias": (E,), "v_proj_1.2.weight": (E, E), "v_proj_1.2.bias": (E,),
                "ln_q_1.weight": (E,), "ln_q_1.bias": (E,), "ln_k_1.weight": (E,), "ln_k_1.bias": (E,),
                "ln_v_1.weight": (E,), "ln_v_1.bias": (E,),
                "clip_attn.in_proj_weight": (3 * E, E), "clip_attn.in_proj_bias": (3 * E,),
                "clip_attn.out_proj.weight": (E, E), "clip_attn.out_proj.bias": (E,),
                "mlp.0.weight": (D, E), "mlp.0.bias": (D,), "mlp.2.weight": (D, D), "mlp.2.bias": (D,)}

    def _ln_eps(self) -> float:
        return float(self.ln_q_1.eps)

    def invalidate_packed(self) -> None:
        """Drop the kernel-side weight image; the next forward re-packs from the current parameters.

        The inference (``no_grad``) path caches the image and re-validates it by ``(data_ptr, _version)`` of every
        parameter, which sees ``optimizer.step()``, ``load_state_dict`` and ``.to()`` but NOT writes that bypass
        autograd's version counter — ``p.data.copy_(...)``, DeepSpeed ZeRO's flat-buffer updates, fused multi-tensor
        optimizers writing through raw pointers.  Every grad-enabled forward therefore re-packs unconditionally and
        leaves the cache invalid behind it; code that rewrites weights out of band between two ``no_grad`` forwards
        calls this."""
        self._packed, self._packed_key = None, None

    def _ensure_packed(self, dtype: torch.dtype, device: torch.device, stream_ptr: int, force: bool = False) -> torch.Tensor:
        """(Re)build the kernel-side weight image: always when ``force`` (training forward), otherwise when any
        parameter storage / version or the compute dtype changed."""
        weights = self._named_weights()
        # (an inference image carries every folded / pre-multiplied weight whatever the tuning table says: the schedule is
        # chosen per forward, the image never has to follow a knob)
        key = (dtype, device, tuple((w.data_ptr(), w._version) for w in weights))
        if not force and self._packed is not None and self._packed_key == key:
            if self._packed_stream != stream_ptr and self._packed_event is not None:
                torch.cuda.current_stream(device).wait_event(self._packed_event)     # packed on another stream: order this one behind it
            return self._packed
        expected = self._expected_shapes()
        for name, w in zip(_capi.WEIGHT_FIELDS, weights):
            if w.numel() == 0 or tuple(w.shape) != expected[name]:
                # DeepSpeed ZeRO-3 partitions every parameter and re-assembles it in per-submodule forward hooks; this module never
                # CALLS its child nn.Linear / nn.LayerNorm containers (they only hold the reference's state-dict names), so those
                # hooks never fire and the kernels would be handed the empty placeholders.  ZeRO-2 (every shipped script of the
                # reference: scripts/v1_5/*.sh -> zero2.json) keeps whole parameters and works.
                raise RuntimeError(
                    f"parameter {name} has shape {tuple(w.shape)} (expected {expected[name]}): the parameter looks partitioned "
                    f"(DeepSpeed ZeRO-3?).  TokenPacker reads its weights directly and does not trigger per-submodule gather hooks; "
                    f"use ZeRO-2, or wrap the forward in deepspeed.zero.GatheredParameters(list(projector.parameters())).")
        stream = torch.cuda.current_stream(device)
        for name, w in zip(_capi.WEIGHT_FIELDS, weights):
            if w.device != device or not w.dtype.is_floating_point:
                raise TypeError(f"parameter {name} is {w.dtype} on {w.device}; inputs are {dtype} on {device} "
                                f"(move the module with .to(...) like the reference does)")
            if w.dtype != dtype and not (torch.is_autocast_enabled("cuda") or self.fp32_compute_dtype is not None):
                raise TypeError(f"parameter {name} is {w.dtype}, inputs are {dtype}: cast the module with .to({dtype}) like "
                                f"the reference does, or run under torch.autocast")
        lib = _capi.load_library()
        desc = _capi.make_desc(1, self.raw_grid, self.scale_factor, self.hidden_size, _DTYPES[dtype],
                               ln_eps=self._ln_eps(), flags=_capi.TP_DESC_TRAIN_PACK if force else 0, tuning=self.tuning)
        nbytes = lib.tp_packed_weight_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise RuntimeError(f"tp_packed_weight_bytes: {_capi.last_error()}")
        packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        # the library keeps a host-side note per image ADDRESS; when this tensor dies its address goes back to the caching
        # allocator, so the note goes too (tp_pack_forget) — a later tensor at the same address is not judged by it
        weakref.finalize(packed, _forget_packed, packed.data_ptr()).atexit = False    # (never during interpreter teardown)
        # fp32 master weights (autocast / fp32_compute_dtype) are rounded to the compute dtype here, like autocast does
        contiguous = [w.detach().to(dtype).contiguous() for w in weights]      # keeps temporaries alive until enqueued
        raw = _capi.tp_weights(*[t.data_ptr() for t in contiguous])
        _capi.check(lib.tp_pack_weights(ctypes.byref(desc), ctypes.byref(raw), packed.data_ptr(), nbytes,
                                        stream_ptr), "tp_pack_weights")
        if not force or not self._overflow_checked:
            # weights beyond the fp16 range (a bf16 model can hold them) were clamped by the pack kernels: refuse
            # instead of silently differing from the reference.  One 4-byte read-back per (rare) inference re-pack
            # and on the first training pack only — a training step must not synchronise.
            off = lib.tp_packed_status_offset(ctypes.byref(desc))
            clamped = int(packed[off:off + 4].view(torch.int32).item())
            self._overflow_checked = True
            if clamped:
                raise OverflowError(f"{clamped} weight element(s) exceed the fp16 range (|w| > 65504) the HIP kernels keep "
                                    f"post-first-layer weights in; the reference would compute with them in {dtype}")
        ev = torch.cuda.Event()
        ev.record(stream)
        self._packed_event, self._packed_stream = ev, stream_ptr
        if force:
            self._packed, self._packed_key = None, None       # a training step's image is never reused (see invalidate_packed)
        else:
            self._packed, self._packed_key = packed, key
        return packed

    _MAX_WORKSPACES = 8          # (device, stream) pairs that keep a workspace; least recently used first out

    def _workspace(self, nbytes: int, device: torch.device, stream_ptr) -> torch.Tensor:
        key = (device.index if device.index is not None else -1, stream_ptr)
        ws = self._workspaces.pop(key, None)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            ws[:_capi.TP_WORKSPACE_STATUS_BYTES].zero_()      # the status block (sticky saturation bits) starts clear
        self._workspaces[key] = ws                           # (re-inserted: dict order = recency)
        while len(self._workspaces) > self._MAX_WORKSPACES:
            self._workspaces.pop(next(iter(self._workspaces)))
        return ws

    def release_stream(self, stream: Optional["torch.cuda.Stream"] = None) -> None:
        """Drop everything kept for ``stream`` (default: the current one): this module's workspace for it and the
        library's side stream (``tp_release_stream``).  Call before destroying a stream the module has run on — the
        runtime may hand the same handle to a new stream."""
        stream = stream if stream is not None else torch.cuda.current_stream()
        ptr = stream.cuda_stream
        for key in [k for k in self._workspaces if k[1] == ptr or k[1] == ("bwd", ptr)]:
            self._workspaces.pop(key)
        with torch.cuda.device(stream.device):
            _capi.check(_capi.load_library().tp_release_stream(ptr), "tp_release_stream")

    # ------------------------------------------------------------------------------------------
    def out_like(self, x: torch.Tensor):
        """``(tensor carrying the result's dtype / device, (M, D))`` for inputs like ``x`` — lets a caller allocate the
        buffer it passes as ``_out``."""
        dt = x.dtype                                         # forward()'s own resolution of the compute / output dtype
        if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") in _DTYPES \
                and x.dtype in (torch.float32, torch.bfloat16, torch.float16):
            dt = torch.get_autocast_dtype("cuda")
        elif x.dtype == torch.float32 and self.fp32_compute_dtype is not None:
            dt = torch.float32
        if self.output_fp32:
            dt = torch.float32
        return torch.empty(0, dtype=dt, device=x.device), (self.num_queries, self.hidden_size)

    def forward(self, x, attn_mask=None, _stage_events=None, _out=None):
        x_multi = x[1]      # multi-level [B, N, 4096] — or its four [B, N, 1024] sources (tokenpacker_amd.tower)
        x = x[0]            # single-level [B, N, 1024]
        parts = None
        if isinstance(x_multi, (list, tuple)):
            parts = tuple(x_multi)
            if len(parts) != 4:
                raise ValueError("x_multi given as parts must be the 4 hidden-state slices (clip_encoder.py:28)")
            x_multi = parts[0]
        if not (x.is_cuda and all(t.is_cuda for t in (parts or (x_multi,)))):
            raise RuntimeError("tokenpacker_amd.TokenPacker runs only on an AMD GPU (HIP kernels); "
                               "there is no CPU fallback")
        if any(t.dtype != x.dtype for t in (parts or (x_multi,))):
            raise TypeError(f"x and x_multi must share a dtype (got {x.dtype}, {x_multi.dtype})")
        compute_dtype, fp32_caller = x.dtype, False
        if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") in _DTYPES \
                and x.dtype in (torch.float32, torch.bfloat16, torch.float16):
            compute_dtype = torch.get_autocast_dtype("cuda")          # what every nn.Linear of the reference would run in
        elif x.dtype == torch.float32 and self.fp32_compute_dtype is not None:
            if self.fp32_compute_dtype not in _DTYPES:
                raise TypeError("fp32_compute_dtype must be torch.float16 or torch.bfloat16")
            compute_dtype, fp32_caller = self.fp32_compute_dtype, True
        if compute_dtype not in _DTYPES:
            raise TypeError(f"supported dtypes: bfloat16 / float16 for both inputs (got {x.dtype}); an fp32 module runs "
                            f"under torch.autocast, or with module.fp32_compute_dtype set (see its docstring)")
        N = self.raw_grid * self.raw_grid
        cm = self.MULTI_LEVEL_DIM // 4 if parts else self.MULTI_LEVEL_DIM
        if x.dim() != 3 or x.shape[1] != N or x.shape[2] != self.embed_dim \
                or any(t.dim() != 3 or tuple(t.shape) != (x.shape[0], N, cm) for t in (parts or (x_multi,))):
            raise ValueError(f"expected x [B,{N},{self.embed_dim}] and x_multi [B,{N},{self.MULTI_LEVEL_DIM}] "
                             f"(or four [B,{N},{self.MULTI_LEVEL_DIM // 4}] parts), "
                             f"got {tuple(x.shape)} and {[tuple(t.shape) for t in (parts or (x_multi,))]}")
        if torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in (parts or (x_multi,)))):
            raise NotImplementedError(
                "the HIP projector does not differentiate with respect to the CLIP features (the reference's tower is "
                "frozen and runs under no_grad, clip_encoder.py:46); detach them")
        mask = self._attn_mask_operand(attn_mask, x.shape[0], x.device) if attn_mask is not None else None
        if mask is not None and (parts or _stage_events is not None):
            raise NotImplementedError("attn_mask takes the concatenated x_multi and no staged timing")
        if x.shape[0] == 0:                  # empty batch: the reference returns an empty [0, M, D] tensor
            out_dtype = torch.float32 if self.output_fp32 else x.dtype
            y = x.new_zeros((0, self.num_queries, self.hidden_size), dtype=out_dtype) if _out is None else _out
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                y = y + 0.0 * sum(p.sum() for p in self.parameters() if p.requires_grad).to(out_dtype)
            return y
        if x.dtype != compute_dtype:                  # autocast / fp32 caller: operands rounded once, like autocast's casts
            x = x.to(compute_dtype)
            if parts:
                parts = tuple(t.to(compute_dtype) for t in parts)
            else:
                x_multi = x_multi.to(compute_dtype)
        # the kernels take element strides (tower outputs are [:,1:] slices); only fix layouts they cannot address
        x = self._addressable(x)
        if parts:
            parts = tuple(self._addressable(t) for t in parts)
            if any(t.stride() != parts[0].stride() for t in parts):       # one stride triple serves all four
                parts = tuple(t.contiguous() for t in parts)
            x_multi = parts
        else:
            x_multi = self._addressable(x_multi)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if _stage_events is not None or self.output_fp32 or fp32_caller or _out is not None or mask is not None:
                raise NotImplementedError("staged timing / fp32 output / fp32_compute_dtype / _out / attn_mask are inference-only")
            # (partially frozen projectors — e.g. LoRA-style experiments freezing the K/V branches — go through the same
            # node: tp_backward computes every gradient, autograd drops those of parameters that do not require one)
            xm = x_multi if parts else (x_multi,)
            return _ProjectFn.apply(self, x, len(xm), *xm, *self._named_weights())
        return self._launch_forward(x, x_multi, train=False, _stage_events=_stage_events, fp32_out=fp32_caller, out=_out,
                                    mask=mask)[0]

    def _attn_mask_operand(self, attn_mask: torch.Tensor, B: int, device):
        """``attn_mask`` as ``nn.MultiheadAttention`` receives it from the reference's forward (builder.py:107,130): 2-D
        ``[1, s*s]`` or 3-D ``[(M*B)*num_heads, 1, s*s]`` (batch index = region * B + image, divide_feature's order),
        boolean (True = masked out) or additive float -> ``(fp32 additive tensor on the device, mask_mode)``."""
        S2 = self.scale_factor ** 2
        m = attn_mask
        if m.dim() == 2 and tuple(m.shape) == (1, S2):
            mode = 1
        elif m.dim() == 3 and tuple(m.shape) == (self.num_queries * B * self.num_heads, 1, S2):
            mode = 2
        else:
            raise ValueError(f"attn_mask must be [1, {S2}] or [{self.num_queries} * B * {self.num_heads}, 1, {S2}] "
                             f"(nn.MultiheadAttention with L = 1, S = {S2}); got {tuple(m.shape)}")
        if m.dtype == torch.bool:
            m = torch.zeros(m.shape, dtype=torch.float32, device=m.device).masked_fill(m, float("-inf"))
        elif not m.dtype.is_floating_point:
            raise TypeError("attn_mask must be boolean or floating point")
        return m.to(device=device, dtype=torch.float32).contiguous(), mode

    # ------------------------------------------------------------------------------------------
    def _launch_forward(self, x, x_multi, train: bool, _stage_events=None, fp32_out: bool = False, out=None, mask=None):
        B, device = x.shape[0], x.device
        parts = x_multi if isinstance(x_multi, tuple) else None
        if parts:
            if _stage_events is not None:
                raise NotImplementedError("staged timing takes the concatenated x_multi")
            part_ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in parts])
            part_strides = _capi.strides3(parts[0].stride())
        # (host cost matters: a one-image forward is bound by the host's enqueue rate — 72 us inside the library's 11 launches, and
        # what this wrapper adds on top, profiles/r06p_timeline_B1.txt: no device-guard when the device is current, ONE
        # current_stream lookup, plain-dict writes instead of nn.Module.__setattr__)
        with (_NULLCTX if device.index == torch.cuda.current_device() else torch.cuda.device(device)):
            stream = torch.cuda.current_stream(device)
            stream_ptr = stream.cuda_stream
            lib = _capi.load_library()
            packed = self._ensure_packed(x.dtype, device, stream_ptr, force=train)
            fp32_out = fp32_out or self.output_fp32
            out_dtype = torch.float32 if fp32_out else x.dtype
            desc = _capi.make_desc(B, self.raw_grid, self.scale_factor, self.hidden_size, _DTYPES[x.dtype],
                                   _capi.TP_F32 if fp32_out else _DTYPES[x.dtype], self._ln_eps(), tuning=self.tuning)
            if out is None:
                out = torch.empty(B, self.num_queries, self.hidden_size, dtype=out_dtype, device=device)
            elif tuple(out.shape) != (B, self.num_queries, self.hidden_size) or out.dtype != out_dtype \
                    or out.device != device or not out.is_contiguous() or out.data_ptr() % 16:
                raise ValueError(f"_out must be a contiguous, 16-byte aligned [{B}, {self.num_queries}, {self.hidden_size}] "
                                 f"{out_dtype} tensor on {device}")
            if train:
                ws_bytes = lib.tp_train_workspace_bytes(ctypes.byref(desc))
                if ws_bytes == 0:
                    raise RuntimeError(f"tp_train_workspace_bytes: {_capi.last_error()}")
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)     # lives until backward has run
                if parts:
                    _capi.check(lib.tp_forward_train_parts(ctypes.byref(desc), x.data_ptr(), _capi.strides3(x.stride()),
                                                           part_ptrs, part_strides, packed.data_ptr(), out.data_ptr(),
                                                           ws.data_ptr(), ws.numel(), stream_ptr), "tp_forward_train_parts")
                else:
                    _capi.check(lib.tp_forward_train(ctypes.byref(desc),
                                                     x.data_ptr(), _capi.strides3(x.stride()),
                                                     x_multi.data_ptr(), _capi.strides3(x_multi.stride()),
                                                     packed.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     stream_ptr), "tp_forward_train")
                return out, ws, desc, packed
            if mask is not None:                 # a masked forward runs the plain schedule (K | V written): larger workspace
                desc.flags |= _capi.TP_DESC_MASKED
            ws_bytes = lib.tp_workspace_bytes(ctypes.byref(desc))
            if ws_bytes == 0:
                raise RuntimeError(f"tp_workspace_bytes: {_capi.last_error()}")
            ws = self._workspace(ws_bytes, device, stream_ptr)
            if parts:
                _capi.check(lib.tp_forward_parts(ctypes.byref(desc), x.data_ptr(), _capi.strides3(x.stride()),
                                                 part_ptrs, part_strides, packed.data_ptr(), out.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), stream_ptr), "tp_forward_parts")
            elif mask is not None:
                _capi.check(lib.tp_forward_masked(ctypes.byref(desc), x.data_ptr(), _capi.strides3(x.stride()),
                                                  x_multi.data_ptr(), _capi.strides3(x_multi.stride()), packed.data_ptr(),
                                                  out.data_ptr(), ws.data_ptr(), ws.numel(), mask[0].data_ptr(), mask[1],
                                                  stream_ptr), "tp_forward_masked")
            elif _stage_events is None:
                _capi.check(lib.tp_forward(ctypes.byref(desc),
                                           x.data_ptr(), _capi.strides3(x.stride()),
                                           x_multi.data_ptr(), _capi.strides3(x_multi.stride()),
                                           packed.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                           stream_ptr), "tp_forward")
            else:
                handles = (ctypes.c_void_p * len(_stage_events))(*[ev.cuda_event for ev in _stage_events])
                _capi.check(lib.tp_forward_staged(ctypes.byref(desc),
                                                  x.data_ptr(), _capi.strides3(x.stride()),
                                                  x_multi.data_ptr(), _capi.strides3(x_multi.stride()),
                                                  packed.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  stream_ptr, handles, len(_stage_events)), "tp_forward_staged")
            self.__dict__["_last_launch"] = (desc, ws, stream_ptr)
            self._poll_saturation(ws, stream)
        return out, ws, desc, packed

    def saturated_stages(self, clear: bool = False):
        """Names of the stages whose fp16 epilogues have CLAMPED a value (where the reference's half-precision arithmetic
        would have produced inf) in any inference forward on the current stream's workspace since the flag was last
        cleared — the sticky status word the kernels maintain (include/tokenpacker.h, TP_WORKSPACE_STATUS_BYTES).  Empty
        tuple = every forward so far stayed inside the fp16 range.  Synchronises (one 4-byte read-back)."""
        if self._last_launch is None:
            return ()
        _, ws, _ = self._last_launch
        bits = int(ws[:4].view(torch.int32).item())
        if clear and bits:
            ws[:4].zero_()
        names = ("query_side",) + tuple(_capi.STAGE_NAMES)
        return tuple(n for i, n in enumerate(names) if bits >> i & 1)

    def _poll_saturation(self, ws: torch.Tensor, stream) -> None:
        """Warn ONCE when a forward clamps: after forwards 1, 2, 4, 8, ... (then every 1024th) the status word is copied
        to pinned host memory behind the forward (asynchronously); whichever later forward finds the copy finished
        reads it.  No synchronisation is ever added to the forward."""
        if self._sat_warned or torch.cuda.is_current_stream_capturing():      # (nothing host-visible inside a graph capture)
            return
        pend = self._sat_pending
        if pend is not None and pend[1].query():
            if int(pend[0].item()) != 0:
                self._sat_warned = True
                import warnings
                warnings.warn("tokenpacker_amd.TokenPacker: an fp16 epilogue saturated (|value| >= 65520 clamped to 65504 "
                              "where the reference would have produced inf); module.saturated_stages() names the stage",
                              RuntimeWarning, stacklevel=3)
                return
            self.__dict__["_sat_pending"] = pend = None
        n = self.__dict__["_sat_count"] = self._sat_count + 1
        if pend is None and ((n & (n - 1)) == 0 or n % 1024 == 0):
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(ws[:4].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            self._sat_pending = (host, ev)

    def saturation_report(self) -> Dict[str, int]:
        """Debug aid: after an inference forward, how many elements of each fp16 intermediate (``q0, Hkv, H2, KV, Q1pre,
        Q, O, A1, A2``) sit at the fp16 clamp bound (or are NaN).  Every epilogue clamps to +-65504 instead of
        producing inf; all zeros means no activation of that forward left the fp16 range.
        Scans the forward's own workspace (``tp_debug_count_saturated``) and synchronises — not for hot loops."""
        if self._last_launch is None:
            raise RuntimeError("saturation_report() needs a preceding inference forward")
        desc, ws, stream_ptr = self._last_launch
        lib = _capi.load_library()
        with torch.cuda.device(ws.device):
            counts = torch.zeros(_capi.TP_NUM_DEBUG_BUFFERS, dtype=torch.int32, device=ws.device)
            _capi.check(lib.tp_debug_count_saturated(ctypes.byref(desc), ws.data_ptr(), ws.numel(), counts.data_ptr(),
                                                     torch.cuda.current_stream(ws.device).cuda_stream),
                        "tp_debug_count_saturated")
            return dict(zip(_capi.DEBUG_BUFFER_NAMES, counts.tolist()))

    def _launch_backward(self, desc, x_multi, train_ws, packed, params, dy):
        parts = tuple(x_multi) if isinstance(x_multi, (tuple, list)) else None
        if parts:
            x_multi = parts[0]
        device = x_multi.device
        with torch.cuda.device(device):
            stream_ptr = torch.cuda.current_stream(device).cuda_stream
            lib = _capi.load_library()
            # `packed` is the image the forward of THIS step ran on; fp32 master weights (autocast) are rounded to the
            # compute dtype for the dgrad operands exactly as they were for the forward
            contiguous = [p.detach().to(x_multi.dtype).contiguous() for p in params]
            raw = _capi.tp_weights(*[t.data_ptr() for t in contiguous])
            grads = [torch.empty_like(t) for t in contiguous]
            gptr = _capi.tp_grads(*[t.data_ptr() for t in grads])
            bw_bytes = lib.tp_backward_workspace_bytes(ctypes.byref(desc))
            if bw_bytes == 0:
                raise RuntimeError(f"tp_backward_workspace_bytes: {_capi.last_error()}")
            bw = self._workspace(bw_bytes, device, ("bwd", stream_ptr))
            dy = dy.to(x_multi.dtype).contiguous()
            if parts:
                part_ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in parts])
                _capi.check(lib.tp_backward_parts(ctypes.byref(desc), part_ptrs, _capi.strides3(parts[0].stride()),
                                                  ctypes.byref(raw), packed.data_ptr(), train_ws.data_ptr(), dy.data_ptr(),
                                                  ctypes.byref(gptr), bw.data_ptr(), bw.numel(), stream_ptr),
                            "tp_backward_parts")
            else:
                _capi.check(lib.tp_backward(ctypes.byref(desc), x_multi.data_ptr(), _capi.strides3(x_multi.stride()),
                                            ctypes.byref(raw), packed.data_ptr(), train_ws.data_ptr(), dy.data_ptr(),
                                            ctypes.byref(gptr), bw.data_ptr(), bw.numel(), stream_ptr), "tp_backward")
            self._watch_backward_status(bw, device)
        return [g if g.dtype == p.dtype else g.to(p.dtype) for g, p in zip(grads, params)]

    def backward_saturated(self) -> int:
        """The sticky saturation word of the LAST backward on the current stream's workspace (include/tokenpacker.h: bit 0 a
        non-finite dy or a clamping GEMM epilogue of the fp16 gradient chain, bit 1 the LayerNorm backward, bit 2 the attention
        backward); 0 = every gradient of that call stayed inside fp16's range.  Synchronises."""
        device = next(self.parameters()).device
        ws = self._workspaces.get((device.index if device.index is not None else -1,
                                   ("bwd", torch.cuda.current_stream(device).cuda_stream)))
        return 0 if ws is None else int(ws[:4].view(torch.int32).item())

    def _watch_backward_status(self, bw: torch.Tensor, device) -> None:
        """No synchronisation: the status word of this backward is copied to pinned memory behind it and looked at when a later
        backward finds the copy finished; warns once."""
        if self._bwd_sat_warned or torch.cuda.is_current_stream_capturing():
            return
        pend = self._bwd_sat_pending
        if pend is not None:
            if not pend[1].query():
                return
            if int(pend[0].item()) != 0:
                self._bwd_sat_warned = True
                import warnings
                warnings.warn("tokenpacker_amd.TokenPacker: a backward pass saturated fp16 (status bits "
                              f"{int(pend[0].item()):#x}: 1 = dy not finite / a GEMM epilogue, 2 = LayerNorm backward, 4 = attention backward) — "
                              "its parameter gradients were clamped; _capi.TuningContext(bwd_chain=1) carries gradients in bf16 instead",
                              RuntimeWarning, stacklevel=3)
                return
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(bw[:4].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._bwd_sat_pending = (host, ev)

    def forward_staged(self, x):
        """Forward that also times every kernel of the schedule with HIP events recorded by the
        library on the launch stream (``tp_forward_staged``).  Returns ``(out, events)``; after a
        synchronize, ``events[i].elapsed_time(events[i+1])`` is stage i's duration in ms
        (names: ``_capi.STAGE_NAMES``).  Benchmark / profiling aid."""
        events = [torch.cuda.Event(enable_timing=True) for _ in range(_capi.TP_NUM_STAGES + 1)]
        stream = torch.cuda.current_stream(x[0].device)
        for ev in events:            # force creation of the underlying hipEvent_t
            ev.record(stream)
        out = self.forward(x, _stage_events=events)
        return out, events

    @staticmethod
    def _addressable(t: torch.Tensor) -> torch.Tensor:
        ok = (t.stride(2) == 1 and t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0)
        return t if ok else t.contiguous()

    def extra_repr(self) -> str:
        return (f"raw_grid={self.raw_grid}, scale_factor={self.scale_factor}, "
                f"num_queries={self.num_queries}, hidden_size={self.hidden_size}, backend=hip/gfx950")


def build_vision_projector(config, **kwargs):
    """Twin of the reference factory (builder.py:144-145): reads ``hidden_size`` and
    ``scale_factor`` only; ``mm_projector_type`` is ignored there too."""
    return TokenPacker(hidden_size=config.hidden_size, scale_factor=config.scale_factor)
