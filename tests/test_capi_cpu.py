"""C-ABI library: loads without a GPU, exports every symbol include/tokenpacker.h declares, and
rejects bad descriptors before touching the device.  No compute calls here."""
import ctypes
import os
import re

import pytest

from tokenpacker_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "tokenpacker.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tp_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load_library()
    declared = _header_symbols()
    assert set(declared) == set(_capi.EXPORTED_SYMBOLS), declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tokenpacker.h but not exported"
    assert lib.tp_version() == _capi.TP_ABI_VERSION


def test_struct_layouts_match_header():
    # tp_desc: 6 x int32 + float + int32 + the tuning-context pointer; tp_weights: 23 pointers
    assert ctypes.sizeof(_capi.tp_desc) == 40
    assert ctypes.sizeof(_capi.tp_weights) == 23 * ctypes.sizeof(ctypes.c_void_p)
    assert len(_capi.WEIGHT_FIELDS) == 23
    assert ctypes.sizeof(_capi.tp_linear_args) == 8 * 4 + 3 * 8 + 6 * 8 + 2 * 4 + 8


def test_sizes_and_descriptor_validation():
    lib = _capi.load_library()
    d = _capi.make_desc(256, 24, 2, 4096, _capi.TP_BF16)
    packed = lib.tp_packed_weight_bytes(ctypes.byref(d))
    # 36,722,688 parameters in 2-byte elements, + fp32 biases/colsums + alignment, + the optional
    # out_proj∘mlp[0] fold (W_om [D,1024] fp16) and its pack-time scratch (Wout^T fp16, fp32 product [D,1024])
    # + the per-head transposes of the folded K in-projection the absorbed schedule reads (w_qt, 2 MiB)
    # + Wc = W'·W2 for k and v and W'q·Wq1 for q (the fused LayerNorm chains, 6 MiB)
    # + the centred chain weights (k, v, q, and the k one's per-head transposes: 8 MiB), the triangular factors R of the three
    #   centred layer-2 weights (6 MiB) and the fp64 scratch of their pack-time Householder QR (3 x [1024][1025] + vectors)
    fold = 4096 * 1024 * 2 + 1024 * 1024 * 2 + 4096 * 1024 * 4 + 1024 * 1024 * 2 + 3 * 1024 * 1024 * 2 + 1024 * 1024 * 2   # (+ w_qt_c)
    # + the rows of the centred V chain weight as K-tile pairs hi_t | lo_t ([1024][2048] fp16, 4 MiB: the absorbed schedule's per-head V GEMM
    #   contracts u against the weight AND its fp16 rounding residual, GemmArgs::a_k_dup; round 4: [hi | hi | lo], 6 MiB; since round 5 interleaved K-tile pairs [hi_t | lo_t], 4 MiB)
    fold += 7 * 1024 * 1024 * 2 + 3 * (1024 * 1025 * 8 + 16 * 1024 * 8 + 16 * 8) + 2 * 1024 * 1024 * 2
    assert 36_722_688 * 2 + fold <= packed < 36_722_688 * 2 + fold + 300_000
    ws = lib.tp_workspace_bytes(ctypes.byref(d))
    assert 1.0e9 < ws < 1.3e9            # schedule-aware: the s = 2 default writes neither H2 nor K | V nor Q1pre
    # bad scale factor: the reference's ValueError (builder.py:51-52)
    bad = _capi.make_desc(1, 24, 5, 4096, _capi.TP_BF16)
    assert lib.tp_workspace_bytes(ctypes.byref(bad)) == 0
    assert "scale_factor must be divisible by grid size" in _capi.last_error()
    with pytest.raises(ValueError, match="scale_factor must be divisible by grid size"):
        _capi.check(_capi.TP_ERR_BAD_SCALE, "x")
    for kw in (dict(batch=0), dict(hidden_size=100), dict(dtype=_capi.TP_F32), dict(out_dtype=7)):
        args = dict(batch=1, raw_grid=24, scale_factor=2, hidden_size=4096, dtype=_capi.TP_BF16)
        out_dtype = kw.pop("out_dtype", None)
        args.update(kw)
        dd = _capi.make_desc(args["batch"], args["raw_grid"], args["scale_factor"], args["hidden_size"],
                             args["dtype"], out_dtype)
        assert lib.tp_packed_weight_bytes(ctypes.byref(dd)) == 0, kw
        assert _capi.last_error()


def test_null_arguments_are_rejected_before_launch():
    lib = _capi.load_library()
    d = _capi.make_desc(1, 24, 2, 256, _capi.TP_BF16)
    st = _capi.strides3((576 * 1024, 1024, 1))
    rc = lib.tp_forward(ctypes.byref(d), None, st, None, st, None, None, None, 0, None)
    assert rc == _capi.TP_ERR_INVALID_ARG
    rc = lib.tp_pack_weights(ctypes.byref(d), None, None, 0, None)
    assert rc == _capi.TP_ERR_INVALID_ARG
    a = _capi.tp_linear_args()
    assert lib.tp_linear(ctypes.byref(a), None) == _capi.TP_ERR_INVALID_ARG
    assert lib.tp_set_tuning(99, 0) == _capi.TP_ERR_INVALID_ARG
    assert lib.tp_set_tuning(_capi.TP_TUNE_GEMM_TILE, 0) == _capi.TP_OK


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    with pytest.raises(_capi.TokenPackerLibraryError, match="no CPU fallback"):
        _capi.load_library(str(tmp_path / "libtokenpacker_hip.so"))


def test_train_hd_and_parts_entry_points_reject_bad_arguments():
    """The round's newer entry points validate before they touch the device, like tp_forward does."""
    lib = _capi.load_library()
    d = _capi.make_desc(1, 24, 2, 256, _capi.TP_BF16)
    st = _capi.strides3((576 * 1024, 1024, 1))
    E = _capi.TP_ERR_INVALID_ARG
    assert lib.tp_forward_train(ctypes.byref(d), None, st, None, st, None, None, None, 0, None) == E
    assert lib.tp_forward_parts(ctypes.byref(d), None, st, None, st, None, None, None, 0, None) == E
    assert lib.tp_forward_train_parts(ctypes.byref(d), None, st, None, st, None, None, None, 0, None) == E
    assert lib.tp_backward(ctypes.byref(d), None, st, None, None, None, None, None, None, 0, None) == E
    assert lib.tp_backward_parts(ctypes.byref(d), None, st, None, None, None, None, None, None, 0, None) == E
    assert _capi.last_error()
    # sizes: the training workspace extends the inference one by the two pre-GELU buffers and every slab the backward reads
    # (the inference one carries the K-split partials of a small batch, which the training forward never uses: compare without)
    _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 2)
    try:
        ws, tws, bws = (f(ctypes.byref(d)) for f in (lib.tp_workspace_bytes, lib.tp_train_workspace_bytes,
                                                     lib.tp_backward_workspace_bytes))
    finally:
        _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 0)
    assert 0 < ws < tws and bws > 0
    bad = _capi.make_desc(1, 24, 5, 256, _capi.TP_BF16)
    assert lib.tp_train_workspace_bytes(ctypes.byref(bad)) == 0
    assert lib.tp_backward_workspace_bytes(ctypes.byref(bad)) == 0
    # HD helpers
    assert lib.tp_hd_rows(1, 1, 144) == 145            # one crop: its tokens + '\n', no global view
    assert lib.tp_hd_rows(2, 3, 144) == (6 * 144 + 6) + 145
    assert lib.tp_hd_assemble(None, 1, None, 1, None, None, None, None, 145, 144, 256, _capi.TP_BF16, None) == E
    plan = (_capi.tp_hd_image * 1)(_capi.tp_hd_image(0, 1, 1, 0, 0))
    assert lib.tp_hd_assemble(plan, 0, None, 1, None, None, None, None, 145, 144, 256, _capi.TP_BF16, None) in (E, _capi.TP_OK)
    assert lib.tp_hd_assemble(plan, 1, None, 1, None, None, None, None, 145, 144, 256, _capi.TP_BF16, None) == E
    # plan validation (host-side, before anything is enqueued): overlap, out-of-order, reads / writes past the buffers
    fake = ctypes.c_void_p(4096)                                   # non-NULL, 16-byte aligned; never dereferenced on a rejected plan
    two = (_capi.tp_hd_image * 2)(_capi.tp_hd_image(0, 2, 2, 0, 0), _capi.tp_hd_image(4, 1, 1, 0, 725))   # 2x2 uses 5 crops
    assert lib.tp_hd_assemble(two, 2, fake, 6, None, fake, fake, fake, 2000, 144, 256, _capi.TP_BF16, None) == E
    assert "overlaps" in lib.tp_last_error().decode()
    two = (_capi.tp_hd_image * 2)(_capi.tp_hd_image(0, 2, 2, 0, 0), _capi.tp_hd_image(5, 1, 1, 0, 700))   # rows 0..724 taken
    assert lib.tp_hd_assemble(two, 2, fake, 6, None, fake, fake, fake, 2000, 144, 256, _capi.TP_BF16, None) == E
    two = (_capi.tp_hd_image * 2)(_capi.tp_hd_image(0, 2, 2, 0, 0), _capi.tp_hd_image(5, 1, 1, 0, 900))   # gap 725..899 is fine ...
    assert lib.tp_hd_assemble(two, 2, fake, 5, None, fake, fake, fake, 2000, 144, 256, _capi.TP_BF16, None) == E   # ... but crop 5 of 5 is not
    assert "crops up to 6 of 5" in lib.tp_last_error().decode()
    assert lib.tp_hd_assemble(two, 2, fake, 6, None, fake, fake, fake, 1000, 144, 256, _capi.TP_BF16, None) == E   # rows up to 1045 of 1000
    # status block of the packed weights, debug scan
    assert 0 < lib.tp_packed_status_offset(ctypes.byref(d)) < lib.tp_packed_weight_bytes(ctypes.byref(d))
    assert lib.tp_debug_count_saturated(ctypes.byref(d), None, 0, None, None) == E
    assert lib.tp_hd_slice(None, 100, 100, 1, 1, 336, 336, 0, 0, None, 336, None) == E
    # the test hooks live in libtokenpacker_exp.so (include/tokenpacker_test.h), not in the product library
    tl = _capi.load_test_library()
    assert tl.tp_test_occupy_cus(0, 1, None, None) == E
    assert tl.tp_test_pack_qr(None, None, None, None, None, None, None) == E          # (test hook of the pack-time QR)
    assert tl.tp_test_pack_qr_scratch_bytes() == 1024 * 1025 * 8 + 16 * 1024 * 8 + 16 * 8 + 256
    for sym in _capi.TEST_SYMBOLS:
        assert not hasattr(lib, sym), f"{sym} must not be exported by the product library"
    assert lib.tp_debug_counter(_capi.TP_COUNTER_SIDE_STREAMS) >= 0 and lib.tp_debug_counter(_capi.TP_COUNTER_PAIR_LAUNCHES) >= 0
    assert lib.tp_debug_counter(99) == -1
    # the timing-probe instantiations (garbage results) are not in the product library: its knob refuses them; the exp build takes them
    assert lib.tp_set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 16) == E and "libtokenpacker_exp" in lib.tp_last_error().decode()
    assert lib.tp_set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 0) == 0
    assert tl.tp_set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 16) == 0 and tl.tp_set_tuning(_capi.TP_TUNE_PAIR_DEBUG, 0) == 0
    # the weight-gradient contraction on its own
    assert lib.tp_wgrad_workspace_bytes(1024, 4096) == 16 * 1024 * 4096 * 4
    assert lib.tp_wgrad_workspace_bytes(0, 4096) == 0
    assert lib.tp_wgrad(None, 1024, None, 4096, 0, 0, 4096, 1024, 4096, _capi.TP_BF16, None, _capi.TP_F32, 0, None, 0, None) == E


def test_tuning_keys_match_the_header():
    """Every TP_TUNE_* key of include/tokenpacker.h exists in the binding with the same number, the binding knows a default
    for each, and tp_set_tuning accepts exactly the keys below TP_TUNE_COUNT_."""
    src = open(os.path.join(ROOT, "include", "tokenpacker.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    keys = dict((k, int(v)) for k, v in re.findall(r"\b(TP_TUNE_[A-Z_0-9]+)\s*=\s*(\d+)", src))
    count = keys.pop("TP_TUNE_COUNT_")
    assert len(set(keys.values())) == len(keys) and max(keys.values()) < count
    for name, num in keys.items():
        assert getattr(_capi, name) == num, name
        assert num in _capi._TUNING_DEFAULTS, f"{name}: no default in _capi._TUNING_DEFAULTS"
    lib = _capi.load_library()
    assert lib.tp_set_tuning(count, 0) == _capi.TP_ERR_INVALID_ARG
    assert lib.tp_set_tuning(-1, 0) == _capi.TP_ERR_INVALID_ARG
    for num in keys.values():
        assert lib.tp_set_tuning(num, _capi._TUNING_DEFAULTS[num]) == _capi.TP_OK


def test_workspace_is_schedule_aware_and_an_upper_bound_for_the_tuning_at_call_time():
    """tp_workspace_bytes sizes the slabs the schedule of the moment writes (plan_schedule): the scale_factor-2 default has no
    H2 / K | V / Q1pre slab (1.2 GB less at B = 256); a masked forward (TP_DESC_MASKED), the unfused chain and the separate
    attention kernel need K | V or H2 again; the K-split partials exist only while TP_TUNE_SPLIT_K is not off."""
    lib = _capi.load_library()
    B, N, E = 256, 576, 1024
    slab = 2 * B * N * E * 2                       # one [2][B N, 1024] fp16 slab (H2, or K | V)

    def size(b=B, s=2, flags=0):
        d = _capi.make_desc(b, 24, s, 4096, _capi.TP_BF16, flags=flags)
        return lib.tp_workspace_bytes(ctypes.byref(d))
    base = size()
    a1 = B * 144 * E * 2                           # out_proj's output: the fold is off where K | V are rounded in front of attention
    try:
        assert slab <= size(flags=_capi.TP_DESC_MASKED) - base <= slab + a1 + 4096    # + K | V (+ A1)
        _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, 1)
        assert slab <= size() - base <= slab + a1 + 4096                                 # separate attention kernel: + K | V (+ A1)
        _capi.set_tuning(_capi.TP_TUNE_FUSE_ATTN, 0)
        _capi.set_tuning(_capi.TP_TUNE_FUSE_KV_LN, 0)
        assert size() - base >= 2 * slab                                                 # + H2, + K | V, + Q1pre, (+ A1)
        _capi.set_tuning(_capi.TP_TUNE_FUSE_KV_LN, 1)
        assert size() == base
        # absorbed schedule (s = 3): u (one fp16 value per element) | qt (fp32) = 3 x [B M, 8, 1024] x 2 B instead of K | V, no H2, no logits
        assert size(s=3) < base + 3 * 8 * B * 64 * E * 2
        # K-split partials: only while the knob is on (the default), only for batches of at most 8 images
        partials = 512 * 128 * 128 * 4
        on = size(b=1)
        big_on = size(b=16)
        _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 2)
        assert partials <= on - size(b=1) < partials + 4096
        assert size(b=16) == big_on
        _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 0)
    finally:
        for k, v in _capi._TUNING_DEFAULTS.items():
            _capi.set_tuning(k, v)
    assert size(b=16) > size(b=9) and size() > 15 * size(b=16) // 2


def test_get_tuning_reads_the_library_table():
    lib = _capi.load_library()
    assert lib.tp_get_tuning(99) == -1 and lib.tp_get_tuning(-1) == -1
    for k, v in _capi._TUNING_DEFAULTS.items():
        assert _capi.get_tuning(k) == v, k
    try:
        assert lib.tp_set_tuning(_capi.TP_TUNE_RESERVE_CUS, 3) == _capi.TP_OK        # a direct call, past the wrapper
        assert _capi.get_tuning(_capi.TP_TUNE_RESERVE_CUS) == 3
    finally:
        _capi.set_tuning(_capi.TP_TUNE_RESERVE_CUS, 0)


def test_gather_entry_points_reject_bad_arguments():
    lib = _capi.load_library()
    E = _capi.TP_ERR_INVALID_ARG
    off = ctypes.c_uint64(0)
    assert lib.tp_gather_export(None, None, ctypes.byref(off)) == E
    assert lib.tp_gather_open(None, None) == E
    assert lib.tp_gather_close(None) == E
    assert lib.tp_gather_alloc_flags(None, 256) == E and lib.tp_gather_free_flags(None) == E
    assert lib.tp_gather_sync(None, 2, 0, 0, None, 0, None, 0, None) == E
    fake = ctypes.c_void_p(4096)
    assert lib.tp_gather_sync(fake, 0, 0, 0, None, 0, None, 0, None) == E          # world < 1
    assert lib.tp_gather_sync(fake, 65, 0, 0, None, 0, None, 0, None) == E         # one wave polls at most 64 sources
    assert lib.tp_gather_sync(fake, 2, 2, 0, None, 0, None, 0, None) == E          # rank out of range
    assert lib.tp_gather_push(1, None, fake, 16, None, fake, None, 0) == E
    assert lib.tp_gather_push(0, None, None, 0, None, None, None, 0) == E          # no sequence cell
    assert _capi.TP_IPC_HANDLE_BYTES == 64


def test_tuning_contexts_are_private_copies():
    """ABI 4: tp_tuning_create copies the process-wide table of the moment; setting a context does not touch the table or another
    context; an entry point that takes a tp_desc reads desc->tuning (here: the workspace size follows the context's SPLIT_K, not the
    table's); bad keys / NULL are errors."""
    lib = _capi.load_library()
    a, b = _capi.TuningContext(), _capi.TuningContext(split_k=2)
    try:
        for k, v in _capi._TUNING_DEFAULTS.items():
            assert a.get(k) == v
        assert b.get(_capi.TP_TUNE_SPLIT_K) == 2 and a.get(_capi.TP_TUNE_SPLIT_K) == 0 and _capi.get_tuning(_capi.TP_TUNE_SPLIT_K) == 0
        size = lambda t: lib.tp_workspace_bytes(ctypes.byref(_capi.make_desc(1, 24, 2, 4096, _capi.TP_BF16, tuning=t)))
        partials = 512 * 128 * 128 * 4
        assert size(None) == size(a) and partials <= size(a) - size(b) < partials + 4096
        _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 2)                  # the table changes under the contexts' feet: they do not move
        try:
            assert size(None) == size(b) and size(a) - size(b) >= partials and a.get(_capi.TP_TUNE_SPLIT_K) == 0
            c = _capi.TuningContext()                               # ... and a NEW context starts from the table of the moment
            assert c.get(_capi.TP_TUNE_SPLIT_K) == 2
            c.close()
        finally:
            _capi.set_tuning(_capi.TP_TUNE_SPLIT_K, 0)
        assert lib.tp_tuning_set(a.handle, _capi.TP_TUNE_COUNT, 0) == _capi.TP_ERR_INVALID_ARG
        assert lib.tp_tuning_set(None, 0, 0) == _capi.TP_ERR_INVALID_ARG and lib.tp_tuning_get(None, 0) == -1
    finally:
        a.close(); b.close()
    # a CLOSED context must not silently fall back to the process-wide table: the caller asked for ITS knobs
    import pytest
    with pytest.raises(ValueError, match="closed"):
        _capi.make_desc(1, 24, 2, 4096, _capi.TP_BF16, tuning=a)
    assert _capi.make_desc(1, 24, 2, 4096, _capi.TP_BF16, tuning=None).tuning is None      # (None stays "the table")


def test_gemm_routing_policy_is_host_logic_and_pinned():
    """Where tp_linear sends a launch is decided on the host by counting CU rounds (tp_gemm.hip: gemm_route / gemm_takes_pair_route);
    tp_test_gemm_route exposes the decision (256 CUs are assumed without a device).  The B = 256 forward runs full 256 x 256 tiles
    (its query-side GEMMs the pair kernel); a 32-image shard's first layer and mlp launches take the 192 x 256 tiles that make them
    whole rounds; one image runs on the 128-tile kernel; statistics-only launches never take the 192-row tiles (not built for them)."""
    lib = _capi.load_test_library()                      # (tp_test_gemm_route: include/tokenpacker_test.h; the exp library's OWN tuning table)
    set_tuning = lambda k, v: lib.tp_set_tuning(k, v)     # noqa: E731
    G, S, NS = _capi.TP_LINEAR_GELU, _capi.TP_LINEAR_ROW_STATS, _capi.TP_LINEAR_NO_STORE
    SMALL, FULL, HALF, SPLIT, T192, PAIR = range(6)
    r = lambda M, N, K, flags=0, groups=1: lib.tp_test_gemm_route(M, N, K, flags, groups)  # noqa: E731
    assert r(0, 256, 256) == -1 and r(256, 100, 256) == -1
    # B = 256 (147456 key rows, 36864 query rows)
    assert r(147456, 2048, 4096, G) == FULL and r(36864, 4096, 4096) == FULL and r(36864, 4096, 1024, G) == FULL
    assert r(147456, 1024, 1024, S | NS, 2) == FULL
    assert r(36864, 1024, 1024) == PAIR
    # the 8-GPU shard: 32 images
    assert r(18432, 2048, 4096, G) == T192 and r(4608, 4096, 4096) == T192 and r(4608, 4096, 1024, G) == T192
    assert r(18432, 1024, 1024, S | NS, 2) != T192
    # one image
    assert r(576, 2048, 4096, G) == SMALL and r(144, 4096, 4096) == SMALL
    # the A/B switches
    set_tuning(_capi.TP_TUNE_GEMM_TILE, 4)
    try:
        assert r(18432, 2048, 4096, G) == SPLIT and r(4608, 4096, 4096) in (SPLIT, HALF)
    finally:
        set_tuning(_capi.TP_TUNE_GEMM_TILE, 0)
    set_tuning(_capi.TP_TUNE_PAIR_GEMM, 1)
    try:
        assert r(36864, 1024, 1024) != PAIR
    finally:
        set_tuning(_capi.TP_TUNE_PAIR_GEMM, 0)
    set_tuning(_capi.TP_TUNE_GEMM_TILE, 3)
    try:
        assert r(147456, 2048, 4096, G) == T192 and r(147456, 1024, 1024, S | NS, 2) != T192
    finally:
        set_tuning(_capi.TP_TUNE_GEMM_TILE, 0)
