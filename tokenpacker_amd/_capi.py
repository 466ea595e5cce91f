"""ctypes binding of ``libtokenpacker_hip.so`` (C ABI declared in ``include/tokenpacker.h``).

The library is the product's only compute path.  There is NO CPU / eager fallback: if the
shared object is missing or a call fails, this module raises (``TokenPackerLibraryError`` /
``RuntimeError`` / ``ValueError``) instead of computing the result some other way.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_uint32, c_uint64, c_void_p)

TP_ABI_VERSION = 5
TP_BF16, TP_F16, TP_F32 = 0, 1, 2
TP_OK, TP_ERR_INVALID_ARG, TP_ERR_BAD_SCALE, TP_ERR_WORKSPACE, TP_ERR_LAUNCH = 0, -1, -2, -3, -4
TP_LINEAR_GELU, TP_LINEAR_LN_FOLD, TP_LINEAR_ROW_STATS = 1, 2, 4
TP_LINEAR_NO_STORE = 64
TP_TUNE_GEMM_TILE, TP_TUNE_XCD_SWIZZLE, TP_TUNE_FOLD_OUT_PROJ, TP_TUNE_DYNAMIC_TILES, TP_TUNE_Q_SIDE_STREAM = 0, 1, 2, 3, 4
TP_TUNE_RESERVE_CUS, TP_TUNE_ABSORB_KV, TP_TUNE_FUSE_KV_LN, TP_TUNE_LN_MERGE, TP_TUNE_FUSE_ATTN = 5, 6, 7, 8, 9
TP_TUNE_SPLIT_K, TP_TUNE_SMALL_GEMM_WAVES, TP_TUNE_TRI_STATS = 10, 11, 12
TP_TUNE_PAIR_GEMM, TP_TUNE_PAIR_STAGGER, TP_TUNE_PAIR_DEBUG = 13, 14, 15
TP_TUNE_DECOUPLE_K, TP_TUNE_BWD_CHAIN = 16, 17
TP_TUNE_COUNT = 18
TP_WGRAD_X_TRANSPOSED = 1
TP_NUM_STAGES = 10
TP_NUM_DEBUG_BUFFERS = 9
DEBUG_BUFFER_NAMES = ("q0", "Hkv", "H2", "KV", "Q1pre", "Q", "O", "A1", "A2")
STAGE_NAMES = ("point_queries", "kv_layer0_gelu", "kv_layer2_stats", "kv_inproj_lnfold", "q_proj_1_stats",
               "q_inproj_lnfold", "region_attention", "out_proj", "mlp0_gelu", "mlp2")

# (TP_LIB_VARIANT=<name>: an A/B build of the same sources made by `make -C tokenpacker_amd/csrc variant NAME=<name> DEFS=...` —
# bench.py / the tools under one build or the other on the same box; unset: the product library)
def _lib_name() -> str:
    variant = os.environ.get("TP_LIB_VARIANT")
    if not variant:
        return "libtokenpacker_hip.so"
    import re
    import warnings
    if not re.fullmatch(r"[A-Za-z0-9_]+", variant):      # (no path separators: only libtokenpacker_<name>.so beside this file)
        raise ValueError(f"TP_LIB_VARIANT={variant!r}: letters, digits and '_' only")
    warnings.warn(f"tokenpacker_amd: TP_LIB_VARIANT={variant} — loading the A/B build libtokenpacker_{variant}.so, NOT the product library",
                  RuntimeWarning, stacklevel=2)
    return f"libtokenpacker_{variant}.so"


LIB_NAME = _lib_name()
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

# every symbol include/tokenpacker.h declares
EXPORTED_SYMBOLS = (
    "tp_version", "tp_last_error", "tp_packed_weight_bytes", "tp_workspace_bytes",
    "tp_pack_weights", "tp_pack_forget", "tp_forward", "tp_forward_staged", "tp_point_queries", "tp_region_attention", "tp_linear",
    "tp_ln_finalize", "tp_linear_stats_parts", "tp_set_tuning", "tp_hd_rows", "tp_hd_assemble",
    "tp_train_workspace_bytes", "tp_backward_workspace_bytes", "tp_forward_train", "tp_backward",
    "tp_forward_parts", "tp_forward_train_parts", "tp_backward_parts", "tp_hd_slice",
    "tp_wgrad", "tp_wgrad_workspace_bytes", "tp_packed_status_offset", "tp_debug_count_saturated", "tp_debug_counter",
    "tp_region_attention_absorbed", "tp_forward_masked",
    "tp_get_tuning", "tp_release_stream",
    "tp_tuning_create", "tp_tuning_destroy", "tp_tuning_set", "tp_tuning_get", "tp_gather_alloc_flags", "tp_gather_free_flags", "tp_gather_export", "tp_gather_open", "tp_gather_close", "tp_gather_push", "tp_gather_sync",
)
# include/tokenpacker_test.h: the stateless test hooks — libtokenpacker_exp.so only (load_test_library()), never the product library
TEST_SYMBOLS = ("tp_test_occupy_cus", "tp_test_pair_occupancy", "tp_test_gemm_route", "tp_test_pack_qr", "tp_test_pack_qr_scratch_bytes")
TP_COUNTER_SIDE_STREAMS, TP_COUNTER_PAIR_LAUNCHES = 0, 1
EXP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtokenpacker_exp.so")

# state-dict name -> tp_weights field order (include/tokenpacker.h)
WEIGHT_FIELDS = (
    "q_proj_1.weight",
    "k_proj_1.0.weight", "k_proj_1.0.bias", "k_proj_1.2.weight", "k_proj_1.2.bias",
    "v_proj_1.0.weight", "v_proj_1.0.bias", "v_proj_1.2.weight", "v_proj_1.2.bias",
    "ln_q_1.weight", "ln_q_1.bias", "ln_k_1.weight", "ln_k_1.bias", "ln_v_1.weight", "ln_v_1.bias",
    "clip_attn.in_proj_weight", "clip_attn.in_proj_bias",
    "clip_attn.out_proj.weight", "clip_attn.out_proj.bias",
    "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
)


class TokenPackerLibraryError(RuntimeError):
    pass


class tp_desc(Structure):
    _fields_ = [("batch", c_int32), ("raw_grid", c_int32), ("scale_factor", c_int32),
                ("hidden_size", c_int32), ("dtype", c_int32), ("out_dtype", c_int32),
                ("ln_eps", c_float), ("flags", c_int32), ("tuning", c_void_p)]


class tp_weights(Structure):
    _fields_ = [(name.replace(".", "_"), c_void_p) for name in WEIGHT_FIELDS]


class tp_grads(Structure):
    """Output pointers of tp_backward: one gradient tensor per parameter, same order as tp_weights."""
    _fields_ = [(name.replace(".", "_"), c_void_p) for name in WEIGHT_FIELDS]


class tp_linear_args(Structure):
    _fields_ = [("M", c_int32), ("N", c_int32), ("K", c_int32), ("dtype", c_int32), ("out_dtype", c_int32),
                ("flags", c_int32), ("rows_per_batch", c_int32), ("a_k_dup", c_int32),
                ("a_batch_stride", c_int64), ("lda", c_int64), ("ldc", c_int64),
                ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("C", c_void_p),
                ("row_mean_rstd", c_void_p), ("colsum", c_void_p),
                ("tile", c_int32), ("ldw", c_int32), ("row_stats_out", c_void_p)]


class tp_hd_image(Structure):
    _fields_ = [("first_crop", c_int32), ("h_block", c_int32), ("w_block", c_int32), ("reserved", c_int32),
                ("out_row", c_int64)]


_lib = None


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the C-ABI library and declare its prototypes.  Loading needs no GPU.

    Inside a torch process, ``import torch`` must come first so that the HIP runtime the library
    binds to (SONAME ``libamdhip64.so.7``) is the one torch already loaded — device pointers and
    streams are then shared.  ``tokenpacker_amd.projector`` guarantees that order.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise TokenPackerLibraryError(
            f"{path} not found: build it with `make -C tokenpacker_amd/csrc` "
            f"(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            f"There is no CPU fallback for the TokenPacker projector.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the environment
        raise TokenPackerLibraryError(f"cannot load {path}: {e}") from e

    lib.tp_version.restype = c_int
    lib.tp_version.argtypes = []
    lib.tp_last_error.restype = c_char_p
    lib.tp_last_error.argtypes = []
    lib.tp_packed_weight_bytes.restype = c_size_t
    lib.tp_packed_weight_bytes.argtypes = [POINTER(tp_desc)]
    lib.tp_workspace_bytes.restype = c_size_t
    lib.tp_workspace_bytes.argtypes = [POINTER(tp_desc)]
    lib.tp_packed_status_offset.restype = c_size_t
    lib.tp_packed_status_offset.argtypes = [POINTER(tp_desc)]
    lib.tp_debug_count_saturated.restype = c_int
    lib.tp_debug_count_saturated.argtypes = [POINTER(tp_desc), c_void_p, c_size_t, c_void_p, c_void_p]
    lib.tp_pack_weights.restype = c_int
    lib.tp_pack_weights.argtypes = [POINTER(tp_desc), POINTER(tp_weights), c_void_p, c_size_t, c_void_p]
    lib.tp_pack_forget.restype = c_int
    lib.tp_pack_forget.argtypes = [c_void_p]
    lib.tp_forward.restype = c_int
    lib.tp_forward.argtypes = [POINTER(tp_desc), c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64),
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.tp_forward_masked.restype = c_int
    lib.tp_forward_masked.argtypes = [POINTER(tp_desc), c_void_p, POINTER(c_int64), c_void_p, POINTER(c_int64),
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p]
    lib.tp_forward_staged.restype = c_int
    lib.tp_forward_staged.argtypes = lib.tp_forward.argtypes + [POINTER(c_void_p), c_int]
    lib.tp_point_queries.restype = c_int
    lib.tp_point_queries.argtypes = [POINTER(tp_desc), c_void_p, POINTER(c_int64), c_void_p, c_void_p]
    lib.tp_region_attention.restype = c_int
    lib.tp_region_attention.argtypes = [POINTER(tp_desc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.tp_region_attention_absorbed.restype = c_int
    lib.tp_region_attention_absorbed.argtypes = [POINTER(tp_desc)] + [c_void_p] * 7
    lib.tp_linear.restype = c_int
    lib.tp_linear.argtypes = [POINTER(tp_linear_args), c_void_p]
    lib.tp_ln_finalize.restype = c_int
    lib.tp_ln_finalize.argtypes = [c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p]
    lib.tp_linear_stats_parts.restype = c_int
    lib.tp_linear_stats_parts.argtypes = [POINTER(tp_linear_args)]
    lib.tp_set_tuning.restype = c_int
    lib.tp_set_tuning.argtypes = [c_int, c_int]
    lib.tp_get_tuning.restype = c_int
    lib.tp_get_tuning.argtypes = [c_int]
    lib.tp_tuning_create.restype = c_void_p
    lib.tp_tuning_create.argtypes = []
    lib.tp_tuning_destroy.restype = None
    lib.tp_tuning_destroy.argtypes = [c_void_p]
    lib.tp_tuning_set.restype = c_int
    lib.tp_tuning_set.argtypes = [c_void_p, c_int, c_int]
    lib.tp_tuning_get.restype = c_int
    lib.tp_tuning_get.argtypes = [c_void_p, c_int]
    lib.tp_release_stream.restype = c_int
    lib.tp_release_stream.argtypes = [c_void_p]
    lib.tp_debug_counter.restype = ctypes.c_longlong
    lib.tp_debug_counter.argtypes = [c_int]
    lib.tp_gather_alloc_flags.restype = c_int
    lib.tp_gather_alloc_flags.argtypes = [POINTER(c_void_p), c_size_t]
    lib.tp_gather_free_flags.restype = c_int
    lib.tp_gather_free_flags.argtypes = [c_void_p]
    lib.tp_gather_export.restype = c_int
    lib.tp_gather_export.argtypes = [c_void_p, c_void_p, POINTER(c_uint64)]
    lib.tp_gather_open.restype = c_int
    lib.tp_gather_open.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.tp_gather_close.restype = c_int
    lib.tp_gather_close.argtypes = [c_void_p]
    lib.tp_gather_push.restype = c_int
    lib.tp_gather_push.argtypes = [c_int, POINTER(c_void_p), c_void_p, c_size_t, POINTER(c_void_p), c_void_p, POINTER(c_void_p), c_int]
    lib.tp_gather_sync.restype = c_int
    lib.tp_gather_sync.argtypes = [c_void_p, c_int, c_int, c_uint32, c_void_p, c_uint32, c_void_p, c_int, c_void_p]
    lib.tp_train_workspace_bytes.restype = c_size_t
    lib.tp_train_workspace_bytes.argtypes = [POINTER(tp_desc)]
    lib.tp_backward_workspace_bytes.restype = c_size_t
    lib.tp_backward_workspace_bytes.argtypes = [POINTER(tp_desc)]
    lib.tp_forward_train.restype = c_int
    lib.tp_forward_train.argtypes = lib.tp_forward.argtypes
    lib.tp_backward.restype = c_int
    lib.tp_backward.argtypes = [POINTER(tp_desc), c_void_p, POINTER(c_int64), POINTER(tp_weights), c_void_p, c_void_p,
                                c_void_p, POINTER(tp_grads), c_void_p, c_size_t, c_void_p]
    lib.tp_forward_parts.restype = c_int
    lib.tp_forward_parts.argtypes = [POINTER(tp_desc), c_void_p, POINTER(c_int64), POINTER(c_void_p), POINTER(c_int64),
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.tp_forward_train_parts.restype = c_int
    lib.tp_forward_train_parts.argtypes = lib.tp_forward_parts.argtypes
    lib.tp_backward_parts.restype = c_int
    lib.tp_backward_parts.argtypes = [POINTER(tp_desc), POINTER(c_void_p), POINTER(c_int64), POINTER(tp_weights), c_void_p,
                                      c_void_p, c_void_p, POINTER(tp_grads), c_void_p, c_size_t, c_void_p]
    lib.tp_wgrad_workspace_bytes.restype = c_size_t
    lib.tp_wgrad_workspace_bytes.argtypes = [c_int, c_int]
    lib.tp_wgrad.restype = c_int
    lib.tp_wgrad.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_int, c_int, c_void_p,
                             c_int, c_int, c_void_p, c_size_t, c_void_p]
    lib.tp_hd_slice.restype = c_int
    lib.tp_hd_slice.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]
    lib.tp_hd_rows.restype = c_int64
    lib.tp_hd_rows.argtypes = [c_int, c_int, c_int]
    lib.tp_hd_assemble.restype = c_int
    lib.tp_hd_assemble.argtypes = [POINTER(tp_hd_image), c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int, c_int, c_int, c_void_p]

    if lib.tp_version() != TP_ABI_VERSION:
        raise TokenPackerLibraryError(
            f"{path}: ABI version {lib.tp_version()} != expected {TP_ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


_test_lib = None


def load_test_library(path: str = EXP_LIB_PATH) -> ctypes.CDLL:
    """dlopen ``libtokenpacker_exp.so`` (include/tokenpacker_test.h: the stateless test hooks, the GEMM kernels' timing-probe
    instantiations, csrc/experimental/) beside the product library.  The two libraries share no state: use this one for the
    ``tp_test_*`` hooks only (or, for the probe tools, as the whole library: ``TP_LIB_VARIANT=exp``)."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(path):
        raise TokenPackerLibraryError(f"{path} not found: build it with `make -C tokenpacker_amd/csrc exp`")
    lib = ctypes.CDLL(path)
    lib.tp_last_error.restype = c_char_p
    lib.tp_last_error.argtypes = []
    lib.tp_test_occupy_cus.restype = c_int
    lib.tp_test_occupy_cus.argtypes = [c_int, c_int, c_void_p, c_void_p]
    lib.tp_test_pair_occupancy.restype = c_int
    lib.tp_test_pair_occupancy.argtypes = []
    lib.tp_test_gemm_route.restype = c_int
    lib.tp_test_gemm_route.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.tp_test_pack_qr_scratch_bytes.restype = c_size_t
    lib.tp_test_pack_qr_scratch_bytes.argtypes = []
    lib.tp_test_pack_qr.restype = c_int
    lib.tp_test_pack_qr.argtypes = [c_void_p] * 7
    lib.tp_set_tuning.restype = c_int
    lib.tp_set_tuning.argtypes = [c_int, c_int]
    lib.tp_version.restype = c_int
    if lib.tp_version() != TP_ABI_VERSION:
        raise TokenPackerLibraryError(f"{path}: ABI version {lib.tp_version()} != expected {TP_ABI_VERSION}; rebuild it")
    _test_lib = lib
    return lib


def last_error() -> str:
    return load_library().tp_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    """Map a tp_status to the reference's exception conventions: bad
    scale_factor / arguments -> ValueError (builder.py:51-52), everything else RuntimeError."""
    if rc == TP_OK:
        return
    msg = last_error()
    if rc in (TP_ERR_BAD_SCALE, TP_ERR_INVALID_ARG):
        raise ValueError(msg if rc == TP_ERR_BAD_SCALE else f"{what}: {msg}")
    raise RuntimeError(f"{what} failed ({rc}): {msg}")


TP_DESC_TRAIN_PACK = 1
TP_DESC_MASKED = 2
TP_IPC_HANDLE_BYTES = 64
TP_WORKSPACE_STATUS_BYTES = 256


def make_desc(batch: int, raw_grid: int, scale_factor: int, hidden_size: int, dtype: int,
              out_dtype: int | None = None, ln_eps: float = 1e-6, flags: int = 0, tuning=None) -> tp_desc:
    """`tuning`: a TuningContext (or its raw handle) the call reads its knobs from; None = the process-wide table."""
    handle = tuning
    if isinstance(tuning, TuningContext):
        handle = tuning.handle
        if not handle:
            # a closed context must not silently fall back to the process-wide table: the caller asked for ITS knobs
            raise ValueError("make_desc: the TuningContext has been closed")
    return tp_desc(batch, raw_grid, scale_factor, hidden_size, dtype,
                   dtype if out_dtype is None else out_dtype, ln_eps, flags, handle)


class TuningContext:
    """A private copy of the library's tuning table (``tp_tuning_create``).  Calls whose descriptor names it read their knobs
    from it and from nothing else: another thread's ``set_tuning`` (the process-wide table) or another context cannot change
    the schedule or the low bits of a forward that runs on this one (include/tokenpacker.h, "tuning knobs").

    Threading contract: ``set`` writes a plain int array that in-flight calls read without a lock — mutate a context only while
    no call that names it is being ENQUEUED on another thread (a forward reads its knobs while it is enqueued, not while its
    kernels run); give each serving thread its own context instead of sharing one that is being tuned."""

    def __init__(self, **knobs):
        self.handle = load_library().tp_tuning_create()
        if not self.handle:
            raise MemoryError(last_error())
        for name, value in knobs.items():
            self.set(globals()["TP_TUNE_" + name.upper()], value)

    def set(self, key: int, value: int) -> None:
        check(load_library().tp_tuning_set(self.handle, key, value), "tp_tuning_set")

    def get(self, key: int) -> int:
        v = load_library().tp_tuning_get(self.handle, key)
        if v == -1 and not (0 <= key < TP_TUNE_COUNT):
            raise ValueError(f"tp_tuning_get: {last_error()}")
        return v

    def close(self) -> None:
        if self.handle:
            load_library().tp_tuning_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:        # noqa: interpreter shutdown
            pass


def strides3(st) -> "ctypes.Array":
    return (c_int64 * 3)(int(st[0]), int(st[1]), int(st[2]))


# the library's defaults (tests reset the table to these)
_TUNING_DEFAULTS = {TP_TUNE_GEMM_TILE: 0, TP_TUNE_XCD_SWIZZLE: 1, TP_TUNE_FOLD_OUT_PROJ: 0, TP_TUNE_DYNAMIC_TILES: 1, TP_TUNE_Q_SIDE_STREAM: 1,
                    TP_TUNE_RESERVE_CUS: 0, TP_TUNE_ABSORB_KV: 0, TP_TUNE_FUSE_KV_LN: 1, TP_TUNE_LN_MERGE: 0, TP_TUNE_FUSE_ATTN: 0,
                    TP_TUNE_SPLIT_K: 0, TP_TUNE_SMALL_GEMM_WAVES: 0, TP_TUNE_TRI_STATS: 0, TP_TUNE_PAIR_GEMM: 0, TP_TUNE_PAIR_STAGGER: 100,
                    TP_TUNE_PAIR_DEBUG: 0, TP_TUNE_DECOUPLE_K: 0, TP_TUNE_BWD_CHAIN: 0}


def set_tuning(key: int, value: int) -> None:
    """Process-wide tuning table of the library (benchmarks / tests; NOT reentrant: include/tokenpacker.h)."""
    check(load_library().tp_set_tuning(key, value), "tp_set_tuning")


def get_tuning(key: int) -> int:
    """The library's current value of a tuning key (``tp_get_tuning``: the table itself, not a copy kept on this side —
    another binding or a direct ``tp_set_tuning`` call cannot make it lie)."""
    v = load_library().tp_get_tuning(key)
    if v == -1 and not (0 <= key < TP_TUNE_COUNT):
        raise ValueError(f"tp_get_tuning: {last_error()}")
    return v


__all__ = [n for n in dir() if n.startswith(("TP_", "tp_"))] + [
    "load_library", "load_test_library", "last_error", "check", "make_desc", "strides3", "set_tuning", "get_tuning", "TuningContext",
    "TokenPackerLibraryError", "WEIGHT_FIELDS", "EXPORTED_SYMBOLS", "TEST_SYMBOLS", "LIB_PATH", "EXP_LIB_PATH", "byref"]
