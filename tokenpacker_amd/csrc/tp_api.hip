// tp_api.hip — extern "C" surface of libtokenpacker_hip.so (see include/tokenpacker.h) and the
// launch schedule of the TokenPacker forward on one MI355X.
//
// Forward schedule (reference builder.py:107-137, line numbers per step):
//   1  point_queries     x -> q0                       [B*M,1024]          :117-118
//   2  linear+GELU       x_multi · [Wk0;Wv0]^T -> Hkv  [B*N,2048]          :112,113 (x_multi read ONCE)
//   3  linear x2 (+stats) Hkv[:, g] · W{k,v}2^T -> H2[g] [2][B*N,1024]     :112,113
//   4  linear x2 LN-fold  LN(H2[g]) · Win{k,v}^T -> K,V [2][B*N,1024]      :112,113 LN + :126-130 in-proj
//   5  linear (+stats)    q0 · Wq1^T -> Q1pre           [B*M,1024]         :120
//   6  linear LN-fold     LN(Q1pre) · Winq^T -> Q       [B*M,1024]         :120 LN + :126-130 in-proj
//   7  region_attention   (Q, K, V) -> O                [B*M,1024]         :122-130
//   8  linear             O · Wout^T + b -> A1          [B*M,1024]         :126-130 out_proj
//   9  linear+GELU        A1 · Wm0^T + b -> A2          [B*M,D]            :136
//   10 linear             A2 · Wm2^T + b -> out         [B*M,D]            :136
#include "tp_internal.h"
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

namespace tp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return TP_ERR_LAUNCH;
    }
    return TP_OK;
}

// Tuning: the process-wide table and the per-call contexts (include/tokenpacker.h).  An entry point that takes a tp_desc opens a
// TuningScope on desc->tuning: every tuning() read of that call, on that host thread, comes from the context (a plain array nobody
// else writes while the call runs) — other threads' tp_set_tuning / tp_tuning_set on other contexts cannot reach it.
static std::atomic<int> g_tuning[TP_TUNE_COUNT_] = {{0}, {1}, {0}, {1}, {1}, {0}, {0}, {1}, {0}, {0}, {0}, {0}, {0}, {0}, {100}, {0}, {0}, {0}};
static_assert(TP_TUNE_XCD_SWIZZLE == 1 && TP_TUNE_DYNAMIC_TILES == 3 && TP_TUNE_Q_SIDE_STREAM == 4 && TP_TUNE_FUSE_KV_LN == 7 &&
              TP_TUNE_PAIR_STAGGER == 14 && TP_TUNE_DECOUPLE_K == 16 && TP_TUNE_BWD_CHAIN == 17 && TP_TUNE_COUNT_ == 18, "defaults above are positional");
}  // namespace tp
struct tp_tuning { int v[TP_TUNE_COUNT_]; };
namespace tp {
static thread_local const tp_tuning* tls_tuning = nullptr;
int tuning(int key) {
    if (key < 0 || key >= TP_TUNE_COUNT_) return 0;
    const tp_tuning* c = tls_tuning;
    return c ? c->v[key] : g_tuning[key].load();
}
TuningScope::TuningScope(const tp_desc* d) : prev_(tls_tuning) { tls_tuning = d ? d->tuning : nullptr; }
TuningScope::~TuningScope() { tls_tuning = (const tp_tuning*)prev_; }

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

PackedLayout packed_layout(int D) {
    PackedLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    const size_t E = kEmbed;
    L.w_kv0 = take(2 * E * kMulti * 2);  L.b_kv0 = take(2 * E * 4);
    L.w_kv2 = take(2 * E * E * 2);       L.b_kv2 = take(2 * E * 4);
    L.w_q1 = take(E * E * 2);
    L.w_in_kv = take(2 * E * E * 2);     L.c_in_kv = take(2 * E * 4);  L.b_in_kv = take(2 * E * 4);
    L.w_in_q = take(E * E * 2);          L.c_in_q = take(E * 4);       L.b_in_q = take(E * 4);
    L.w_qt = take(E * E * 2);
    L.w_qt_c = take(E * E * 2);
    L.w_c_kv = take(2 * E * E * 2);      L.d_in_kv = take(2 * E * 4);
    L.w_c_q = take(E * E * 2);
    L.w_cc_kv = take(2 * E * E * 2);     L.d_cc_kv = take(2 * E * 4);
    L.w_cc_q = take(E * E * 2);          L.w_qt_cc = take(E * E * 2);
    L.w_cc_v3 = take(2 * E * E * 2);
    L.w_r_kv = take(2 * E * E * 2);      L.c_r_kv = take(2 * E * 4);
    L.w_r_q = take(E * E * 2);
    L.wbar = take(3 * (E + 1) * 4);      L.scratch_qr = take(pack_qr_scratch_bytes(3));
    L.w_out = take(E * E * 2);           L.b_out = take(E * 4);
    L.w_m0 = take((size_t)D * E * 2);    L.b_m0 = take((size_t)D * 4);
    L.w_m2 = take((size_t)D * D * 2);    L.b_m2 = take((size_t)D * 4);
    L.w_om = take((size_t)D * E * 2);    L.b_om = take((size_t)D * 4);
    L.scratch_t = take(E * E * 2);       L.scratch_p = take((size_t)(D > (int)E ? D : (int)E) * E * 4);   // fp32 [max(D, E), E]
    L.status = take(256);
    L.total = off;
    return L;
}

WorkspaceLayout workspace_layout(int B, int grid, int s, int D, const SchedulePlan& P) {
    WorkspaceLayout L{};
    const size_t N = (size_t)grid * grid, G = grid / s, M = G * G, E = kEmbed;
    const size_t rows_kv = (size_t)B * N, rows_q = (size_t)B * M;
    L.stats_parts_kv = gemm_stats_parts(kEmbed);
    L.stats_parts_q = gemm_stats_parts(kEmbed);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    auto take_if = [&](bool need, size_t bytes) { return need ? take(bytes) : kNoSlab; };
    L.status = take(256);                                   // offset 0 whatever the schedule: the caller zeroes it once
    L.q0 = take(rows_q * E * 2);
    L.hkv = take(rows_kv * 2 * E * 2);
    {
        const size_t lg = kHeads * rows_kv * 4, mru = kHeads * rows_q * 2 * 4;
        L.attn_aux = take(lg > mru ? lg : mru);
    }
    L.h2 = take_if(P.need_h2, 2 * rows_kv * E * 2);
    L.stats_kv = take(2 * (size_t)8 * rows_kv * 2 * 4);     // up to 8 slabs (tile 128) per group
    L.mr_kv = take(2 * rows_kv * 2 * 4);                    // per-row (mean, rstd), 2 groups
    // K | V [2][rows_kv, E] (training, masked / plain schedules); the absorbed schedule keeps qt | u [2][rows_q, 8, E] there
    // (u_split: u [rows_q, 8, E] fp16 FIRST — what the saturation scan looks at —, then qt [rows_q, 8, E] in fp32)
    L.kv = take_if(P.need_kv, P.absorb ? (P.u_split ? 3 : 2) * 8 * rows_q * E * 2 : 2 * rows_kv * E * 2);   // qt | u, or u | qt (fp32)
    L.q1pre = take_if(P.need_q1pre, rows_q * E * 2);
    L.stats_q = take((size_t)8 * rows_q * 2 * 4);
    L.mr_q = take(rows_q * 2 * 4);
    L.q = take(rows_q * E * 2);
    L.o = take(rows_q * E * 2);
    L.a1 = take_if(P.need_a1, rows_q * E * 2);
    L.a2 = take(rows_q * (size_t)D * 2);
    L.counters = take(kCounterBytes);                       // tile-queue heads + stream-K flags of every launch, zeroed per forward
    L.splitk = take_if(P.split_k, kSplitKBytes);
    L.z1 = L.z2 = kNoSlab;
    if (P.train) { L.z1 = take(rows_kv * 2 * E * 2); L.z2 = take(rows_q * (size_t)D * 2); }
    L.total = off;
    return L;
}

WorkspaceLayout workspace_layout(int B, int grid, int s, int D, bool train) {
    tp_desc d{};
    d.batch = B; d.raw_grid = grid; d.scale_factor = s; d.hidden_size = D; d.dtype = TP_BF16; d.out_dtype = TP_BF16; d.ln_eps = 1e-6f;
    return workspace_layout(B, grid, s, D, plan_schedule(&d, train, false));
}

int validate_desc(const tp_desc* d) {
    if (!d) { set_error("tp_desc is NULL"); return TP_ERR_INVALID_ARG; }
    if (d->batch <= 0 || d->raw_grid <= 0 || d->scale_factor <= 0 || d->hidden_size <= 0) {
        set_error("tp_desc: batch/raw_grid/scale_factor/hidden_size must be positive (got %d/%d/%d/%d)",
                  d->batch, d->raw_grid, d->scale_factor, d->hidden_size);
        return TP_ERR_INVALID_ARG;
    }
    if (d->raw_grid % d->scale_factor != 0) {
        set_error("scale_factor must be divisible by grid size");    // the reference's message, builder.py:52
        return TP_ERR_BAD_SCALE;
    }
    if (d->hidden_size % 128 != 0) {
        set_error("tp_desc: hidden_size %d must be a multiple of 128", d->hidden_size);
        return TP_ERR_INVALID_ARG;
    }
    if (d->dtype != TP_BF16 && d->dtype != TP_F16) {
        set_error("tp_desc: dtype %d unsupported (TP_BF16 / TP_F16)", d->dtype);
        return TP_ERR_INVALID_ARG;
    }
    if (d->out_dtype != d->dtype && d->out_dtype != TP_F32) {
        set_error("tp_desc: out_dtype %d must equal dtype or be TP_F32", d->out_dtype);
        return TP_ERR_INVALID_ARG;
    }
    if ((long long)d->batch * d->raw_grid * d->raw_grid > 0x7fffffffLL / 2) {
        set_error("tp_desc: batch*grid*grid too large for 32-bit row indices");
        return TP_ERR_INVALID_ARG;
    }
    if (!(d->ln_eps > 0.f)) { set_error("tp_desc: ln_eps must be > 0"); return TP_ERR_INVALID_ARG; }
    if (d->flags & ~(TP_DESC_TRAIN_PACK | TP_DESC_MASKED)) { set_error("tp_desc: unknown flags 0x%x", d->flags); return TP_ERR_INVALID_ARG; }
    return TP_OK;
}

// The absorbed schedule (region_attention_absorbed_kernel's header): auto = scale_factor >= 3, where the two in-projection
// GEMMs over the B*576 fine tokens cost far more than the 1/s^2-sized query-side work that replaces them (at s = 2 the
// [B M, 8, 1024] intermediates cost what the GEMMs save).  Inference only: the backward needs K and V.
bool absorb_kv(const tp_desc* d, bool train) {
    if (train) return false;
    const int s2 = d->scale_factor * d->scale_factor, mode = tuning(TP_TUNE_ABSORB_KV);
    if (s2 > 64 || mode == 1) return false;
    return mode == 2 || d->scale_factor >= 3;
}

// ONE place decides the schedule of a forward (SchedulePlan, tp_internal.h).
// out_proj folded into mlp[0] (W = Wm0·Wout): TP_TUNE_FOLD_OUT_PROJ 0 = auto, 1 = always, 2 = never.  Auto folds where K and V
// are not rounded to fp16 in front of attention — the absorbed schedule and attention in the in-projections' epilogues: the
// fold removes a 2.25-round launch (-2.3 % of the B = 256 forward) and moves the max-norm parity metric by its own noise
// (worst s = 2 golden case 0.68e-3 without, 0.79e-3 with it, rel-L2 unchanged; tools/fold_parity.py).  It is decided PER
// FORWARD from what actually runs: a masked forward, or a grid whose rows per image are not a multiple of 8, falls back to
// the plain schedule (K, V rounded) and therefore does NOT fold — there the worst case sat at 0.984e-3 / 0.992e-3 of the
// 1e-3 gate.  The folded weight is part of every inference pack, so the choice needs no re-pack.
SchedulePlan plan_schedule(const tp_desc* d, bool train, bool masked) {
    SchedulePlan P{};
    const int s = d->scale_factor, N = d->raw_grid * d->raw_grid;
    const bool chain = tuning(TP_TUNE_FUSE_KV_LN) != 0;
    P.train = train;
    P.absorb = absorb_kv(d, train);
    P.absorb_raw = P.absorb && chain;
    P.fuse_ln = !train && !P.absorb && chain;
    P.fuse_q = !train && chain;
    P.region_major = P.fuse_ln && s == 2 && (N % 8) == 0 && !masked && tuning(TP_TUNE_FUSE_ATTN) != 1;
    P.fuse_attn = P.region_major && tuning(TP_TUNE_FUSE_ATTN) == 0;
    const int fmode = tuning(TP_TUNE_FOLD_OUT_PROJ);
    P.fold = !train && (fmode == 1 || (fmode == 0 && (P.absorb || P.fuse_attn)));
    P.tri = !train && chain && tuning(TP_TUNE_TRI_STATS) != 1;
    P.u_split = P.absorb_raw && P.tri;
    P.decouple_k = P.fuse_attn && P.tri && tuning(TP_TUNE_DECOUPLE_K) != 0;
    P.split_k = tuning(TP_TUNE_SPLIT_K) != 2 && !train && d->batch <= 8;      // (default on since round 3: see the header)
    P.need_h2 = train || !(P.fuse_ln || P.absorb_raw);
    P.need_kv = train || P.absorb || !P.fuse_attn;
    P.need_q1pre = !P.fuse_q;
    P.need_a1 = !P.fold;
    return P;
}

bool fold_out_proj(const tp_desc* d, bool train) { return plan_schedule(d, train, false).fold; }

// Host-side record of what tp_pack_weights put into an image (keyed by device + address; the image itself lives in device
// memory, and a forward must not synchronise to read a header back).  The forward entry points refuse an image that cannot
// serve them — a TRAIN_PACK image handed to an inference entry point, or an image packed for another hidden size / dtype —
// instead of multiplying by weights that were never written.  An address the registry has not seen (an image the caller
// copied elsewhere) is accepted unchecked.  256 entries, least recently packed first out.  The record describes an ADDRESS: a
// caller that frees an image and lets its allocator hand the address to something else (another image memcpy'd there) tells the
// library with tp_pack_forget — the torch binding does so whenever it drops an image — or the stale record answers for it.
struct PackRec { int dev; const void* ptr; int train_pack, D, dtype; unsigned long long stamp; };
static std::mutex g_pack_mu;
static std::vector<PackRec> g_pack_reg;
static unsigned long long g_pack_clock = 0;
void pack_registry_put(const void* packed, const tp_desc* d, bool train_pack) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_pack_mu);
    for (auto& r : g_pack_reg)
        if (r.dev == dev && r.ptr == packed) { r.train_pack = train_pack; r.D = d->hidden_size; r.dtype = d->dtype; r.stamp = ++g_pack_clock; return; }
    if (g_pack_reg.size() >= 256) {
        size_t lru = 0;
        for (size_t i = 1; i < g_pack_reg.size(); ++i) if (g_pack_reg[i].stamp < g_pack_reg[lru].stamp) lru = i;
        g_pack_reg.erase(g_pack_reg.begin() + (long)lru);
    }
    g_pack_reg.push_back(PackRec{dev, packed, train_pack ? 1 : 0, d->hidden_size, d->dtype, ++g_pack_clock});
}
void pack_registry_forget(const void* packed) {
    // Matched on the POINTER alone (device pointers are unique in the process's unified address space): the caller is typically a
    // finalizer that runs wherever the tensor happens to die — the autograd thread, another device current — and a device test here
    // would silently leave the stale record behind.
    std::lock_guard<std::mutex> lock(g_pack_mu);
    for (size_t i = 0; i < g_pack_reg.size();)
        if (g_pack_reg[i].ptr == packed) g_pack_reg.erase(g_pack_reg.begin() + (long)i); else ++i;
}
int pack_registry_check(const void* packed, const tp_desc* d, bool train) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TP_OK;
    std::lock_guard<std::mutex> lock(g_pack_mu);
    for (const auto& r : g_pack_reg)
        if (r.dev == dev && r.ptr == packed) {
            if (r.D != d->hidden_size || r.dtype != d->dtype) {
                set_error("packed weights at %p were packed for hidden_size %d / dtype %d, the forward asks for %d / %d", packed, r.D, r.dtype,
                          d->hidden_size, d->dtype);
                return TP_ERR_INVALID_ARG;
            }
            if (r.train_pack && !train) {
                set_error("packed weights at %p are a TP_DESC_TRAIN_PACK image (no inference-only weights): tp_forward_train only", packed);
                return TP_ERR_INVALID_ARG;
            }
            return TP_OK;
        }
    return TP_OK;
}

GemmArgs plain_gemm(const void* A, long long lda_elems, const void* W, void* C, long long ldc,
                    int M, int N, int K, const float* bias, int flags) {
    GemmArgs a{};
    a.A = (const char*)A; a.W = (const char*)W; a.C = (char*)C;
    a.bias = bias;
    a.lda_bytes = lda_elems * 2; a.a_batch_stride_bytes = 0; a.rows_per_batch = M;
    a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.flags = flags; a.groups = 1;
    return a;
}

}  // namespace tp

using namespace tp;
static constexpr int BK_ELEMS = 64;                  // K-slab of the GEMM kernels (tp_gemm_common.h BK)

extern "C" {

int tp_version(void) { return TP_ABI_VERSION; }

const char* tp_last_error(void) { return g_err; }

int tp_set_tuning(int key, int value) {
    if (key < 0 || key >= TP_TUNE_COUNT_) { set_error("tp_set_tuning: bad key %d", key); return TP_ERR_INVALID_ARG; }
    if (key == TP_TUNE_PAIR_DEBUG && value != 0 && !gemm_probes_built()) { set_error("tp_set_tuning: TP_TUNE_PAIR_DEBUG selects timing-probe kernels that are not in this library (libtokenpacker_exp.so: make exp)"); return TP_ERR_INVALID_ARG; }
    g_tuning[key].store(value);
    return TP_OK;
}

int tp_get_tuning(int key) {
    if (key < 0 || key >= TP_TUNE_COUNT_) { set_error("tp_get_tuning: bad key %d", key); return -1; }
    return g_tuning[key].load();
}

tp_tuning* tp_tuning_create(void) {
    tp_tuning* t = new (std::nothrow) tp_tuning;
    if (!t) { set_error("tp_tuning_create: out of memory"); return nullptr; }
    for (int k = 0; k < TP_TUNE_COUNT_; ++k) t->v[k] = g_tuning[k].load();
    return t;
}
void tp_tuning_destroy(tp_tuning* t) { delete t; }
int tp_tuning_set(tp_tuning* t, int key, int value) {
    if (!t || key < 0 || key >= TP_TUNE_COUNT_) { set_error("tp_tuning_set: bad context / key %d", key); return TP_ERR_INVALID_ARG; }
    if (key == TP_TUNE_PAIR_DEBUG && value != 0 && !gemm_probes_built()) { set_error("tp_tuning_set: TP_TUNE_PAIR_DEBUG selects timing-probe kernels that are not in this library (libtokenpacker_exp.so: make exp)"); return TP_ERR_INVALID_ARG; }
    t->v[key] = value;
    return TP_OK;
}
int tp_tuning_get(const tp_tuning* t, int key) {
    if (!t || key < 0 || key >= TP_TUNE_COUNT_) { set_error("tp_tuning_get: bad context / key %d", key); return -1; }
    return t->v[key];
}

}  // extern "C"
namespace tp {
// Images one forward launch sequence can take: every GEMM output of the path must stay below the 4 GiB a launch addresses
// (range-checked 32-bit store offsets, tp_gemm_common.h) — Hkv / Z1 [B N, 2E] fp16 and A2 / out [B M, D].
long long max_images_per_launch(const tp_desc* desc) {
    const long long n_tok = (long long)desc->raw_grid * desc->raw_grid;
    const long long m_tok = n_tok / ((long long)desc->scale_factor * desc->scale_factor);
    const long long out_esz = desc->out_dtype == TP_F32 ? 4 : 2, lim = 1ll << 32;
    const long long bkv = (lim / (2 * kEmbed * 2) - 256) / n_tok;
    const long long bq = (lim / ((long long)desc->hidden_size * out_esz) - 256) / m_tok;
    return bq < bkv ? bq : bkv;
}
}  // namespace tp
using namespace tp;
extern "C" {

size_t tp_packed_weight_bytes(const tp_desc* desc) {
    TuningScope tuning_scope(desc);
    if (validate_desc(desc) != TP_OK) return 0;
    return packed_layout(desc->hidden_size).total;
}

size_t tp_packed_status_offset(const tp_desc* desc) {
    TuningScope tuning_scope(desc);
    if (validate_desc(desc) != TP_OK) return 0;
    return packed_layout(desc->hidden_size).status;
}

size_t tp_workspace_bytes(const tp_desc* desc) {
    TuningScope tuning_scope(desc);
    if (validate_desc(desc) != TP_OK) return 0;
    const long long bc = max_images_per_launch(desc);    // larger batches run as chunks of this size through one workspace
    const int b = (bc >= 1 && desc->batch > bc) ? (int)bc : desc->batch;
    tp_desc d = *desc; d.batch = b;
    return workspace_layout(b, desc->raw_grid, desc->scale_factor, desc->hidden_size,
                            plan_schedule(&d, false, (desc->flags & TP_DESC_MASKED) != 0)).total;
}

int tp_pack_weights(const tp_desc* desc, const tp_weights* raw, void* packed, size_t packed_bytes,
                    void* stream_) {
    TuningScope tuning_scope(desc);
    int rc = validate_desc(desc);
    if (rc != TP_OK) return rc;
    if (!raw || !packed) { set_error("tp_pack_weights: NULL argument"); return TP_ERR_INVALID_ARG; }
    const void* const* ptrs = reinterpret_cast<const void* const*>(raw);
    for (size_t i = 0; i < sizeof(tp_weights) / sizeof(void*); ++i)
        if (!ptrs[i]) { set_error("tp_pack_weights: weight pointer #%zu is NULL", i); return TP_ERR_INVALID_ARG; }
    const int D = desc->hidden_size, dt = desc->dtype;
    const PackedLayout L = packed_layout(D);
    if (packed_bytes < L.total) {
        set_error("tp_pack_weights: packed buffer %zu B < required %zu B", packed_bytes, L.total);
        return TP_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    char* P = (char*)packed;
    const size_t E = kEmbed;
#define TP_TRY(expr) do { int rc_ = (expr); if (rc_ != TP_OK) return rc_; } while (0)
    int* status = (int*)(P + L.status);
    int* sat = status;                                   // [0]: weight elements clamped to the fp16 range
    {
        hipError_t e = hipMemsetAsync(status, 0, 256, stream);
        if (e != hipSuccess) { set_error("tp_pack_weights: hipMemsetAsync: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    // The casts and copies of the image, gathered into ONE launch (round 6: the training pack runs every step — twenty 4-27 us
    // launches were 0.15 ms of it): K/V first layers concatenated (one GEMM reads x_multi once); everything after the first layer
    // runs on fp16 activations, so those weights are widened to fp16 (exact for in-range bf16 values, identity for fp16 models);
    // biases in fp32.
    BatchOps ops{};
    auto c16 = [&](const void* src, size_t off, long long n) { return ops.add(BATCH_OP_TO_F16, src, P + off, n); };
    auto c32 = [&](const void* src, float* dst, long long n) { return ops.add(BATCH_OP_TO_F32, src, dst, n); };
    bool fits = ops.add(BATCH_OP_COPY16, raw->k_proj_1_0_weight, P + L.w_kv0, (long long)(E * kMulti));
    fits = fits && ops.add(BATCH_OP_COPY16, raw->v_proj_1_0_weight, P + L.w_kv0 + E * kMulti * 2, (long long)(E * kMulti));
    fits = fits && c32(raw->k_proj_1_0_bias, (float*)(P + L.b_kv0), (long long)E) && c32(raw->v_proj_1_0_bias, (float*)(P + L.b_kv0) + E, (long long)E);
    fits = fits && c16(raw->k_proj_1_2_weight, L.w_kv2, (long long)(E * E)) && c16(raw->v_proj_1_2_weight, L.w_kv2 + E * E * 2, (long long)(E * E));
    fits = fits && c32(raw->k_proj_1_2_bias, (float*)(P + L.b_kv2), (long long)E) && c32(raw->v_proj_1_2_bias, (float*)(P + L.b_kv2) + E, (long long)E);
    fits = fits && c16(raw->q_proj_1_weight, L.w_q1, (long long)(E * E));
    fits = fits && c16(raw->clip_attn_out_proj_weight, L.w_out, (long long)(E * E)) && c32(raw->clip_attn_out_proj_bias, (float*)(P + L.b_out), (long long)E);
    fits = fits && c16(raw->mlp_0_weight, L.w_m0, (long long)D * E) && c32(raw->mlp_0_bias, (float*)(P + L.b_m0), (long long)D);
    fits = fits && c16(raw->mlp_2_weight, L.w_m2, (long long)D * D) && c32(raw->mlp_2_bias, (float*)(P + L.b_m2), (long long)D);
    if (!fits) { set_error("tp_pack_weights: batch table too small"); return TP_ERR_LAUNCH; }
    TP_TRY(pack_batch_launch(dt, ops, stream, sat));
    // LayerNorm affines folded into the q/k/v in-projections (in_proj rows: q | k | v)
    const char* inw = (const char*)raw->clip_attn_in_proj_weight;
    const char* inb = (const char*)raw->clip_attn_in_proj_bias;
    TP_TRY(pack_ln_fold_launch(dt, inw, inb, raw->ln_q_1_weight, raw->ln_q_1_bias, P + L.w_in_q,
                               (float*)(P + L.c_in_q), (float*)(P + L.b_in_q), (int)E, (int)E, stream, sat));
    TP_TRY(pack_ln_fold_launch(dt, inw + E * E * 2, inb + E * 2, raw->ln_k_1_weight, raw->ln_k_1_bias,
                               P + L.w_in_kv, (float*)(P + L.c_in_kv), (float*)(P + L.b_in_kv), (int)E, (int)E, stream, sat));
    TP_TRY(pack_ln_fold_launch(dt, inw + 2 * E * E * 2, inb + 2 * E * 2, raw->ln_v_1_weight, raw->ln_v_1_bias,
                               P + L.w_in_kv + E * E * 2, (float*)(P + L.c_in_kv) + E, (float*)(P + L.b_in_kv) + E,
                               (int)E, (int)E, stream, sat));
    const bool train_pack = (desc->flags & TP_DESC_TRAIN_PACK) != 0;     // training image: inference-only weights are skipped
    if (!train_pack) TP_TRY(pack_head_transpose_launch(P + L.w_in_kv, P + L.w_qt, stream));      // of the ROUNDED W'k: the absorbed schedule
    // Fused LayerNorm chain of the K/V side (inference, plain schedule; TP_TUNE_FUSE_KV_LN):
    //   K = LN(H2)·Win^T + b,  H2 = Hkv·W2^T + b2   =>   K = rstd·(Hkv·Wc^T + d − mu·c) + b',
    //   Wc = W'·W2 (fp32 accumulate on the MFMA kernel, rounded once to fp16),  d = W'·b2,  (mu, rstd) = row statistics
    //   of H2 — which is then computed for those statistics only and never written (604 MB less written and read per
    //   B = 256 forward, 16 fewer output stores per tile of that GEMM).  W' is the ROUNDED folded weight, so the mean
    //   term still cancels against c = rowsum(W') exactly as in the unfused form.
    // Every INFERENCE pack carries every inference-only weight, whatever the tuning table says at pack time: the schedule is
    // chosen per forward (plan_schedule) and must find its operands whichever knob was turned in between.
    if (!train_pack) {
        // triangular statistics (tp_pack_qr.hip): W2 / b2 centred over the output index (fp64), factored W2c = Q R; the chain
        // weights of the centred form are built below from the same fp32 products as the plain ones
        float* wbar = (float*)(P + L.wbar);
        for (int g = 0; g < 2; ++g)
            TP_TRY(pack_qr_center_launch(P + L.w_kv2 + (size_t)g * E * E * 2, (const float*)(P + L.b_kv2) + g * E, P + L.scratch_qr, g,
                                         wbar + g * (E + 1), stream));
        TP_TRY(pack_qr_center_launch(P + L.w_q1, nullptr, P + L.scratch_qr, 2, wbar + 2 * (E + 1), stream));
        TP_TRY(pack_qr_factor_launch(P + L.scratch_qr, 3, stream));
        for (int g = 0; g < 2; ++g)
            TP_TRY(pack_qr_extract_launch(P + L.scratch_qr, g, P + L.w_r_kv + (size_t)g * E * E * 2, (float*)(P + L.c_r_kv) + g * E, stream, sat));
        TP_TRY(pack_qr_extract_launch(P + L.scratch_qr, 2, P + L.w_r_q, nullptr, stream, sat));
        // The centred chain weights are built from the UNROUNDED fold (round 4): W'_nk = W_nk gamma_k is exact in fp32, the fp16 W'
        // the unfused schedules keep is its rounding `hi`; with lo = fp16(W' − hi) the product (hi + lo)·W2 carries the fold to
        // ~2^-22, so the fold's rounding (one of the ~6 fp16 roundings in series on the value path) is gone from Wcc / dcc — which
        // need no rowsum of the ROUNDED W' either (no mean term in the centred form).  The QR workspace is dead by now: it holds lo
        // [E, E] fp16, the second product [E, E] fp32 and the exact rowsum / bias fold.
        char* const xs = P + L.scratch_qr;
        char* const lo16 = xs;
        float* const P2 = (float*)(xs + E * E * 2);
        float* const c_ex = (float*)(xs + E * E * 2 + E * E * 4);
        float* const d_ex = c_ex + E;
        static_assert(kEmbed * kEmbed * 6 + 2 * kEmbed * 4 <= 3 * (size_t)kEmbed * (kEmbed + 1) * 8, "QR scratch holds the residual products");
        const void* const gammas[3] = {raw->ln_k_1_weight, raw->ln_v_1_weight, raw->ln_q_1_weight};
        for (int g = 0; g < 2; ++g) {
            TP_TRY(pack_transpose_f16_launch(P + L.w_kv2 + (size_t)g * E * E * 2, P + L.scratch_t, (int)E, stream));   // W2^T [k][j]
            GemmArgs a = plain_gemm(P + L.w_in_kv + (size_t)g * E * E * 2, E, P + L.scratch_t, P + L.scratch_p, E, (int)E, (int)E,
                                    (int)E, nullptr, 0);
            a.tile = 128;
            TP_TRY(gemm_launch(TP_F16, TP_F32, a, stream));                      // Wc[n][k] = sum_j W'[n][j] W2[j][k]
            TP_TRY(pack_round_f16_launch((const float*)(P + L.scratch_p), P + L.w_c_kv + (size_t)g * E * E * 2, (long long)E * E, stream, sat));
            TP_TRY(pack_bias_fold_launch(P + L.w_in_kv + (size_t)g * E * E * 2, (const float*)(P + L.b_kv2) + g * E, nullptr,
                                         (float*)(P + L.d_in_kv) + g * E, (int)E, (int)E, stream));
            // the centred chain: Wc' = W'·W2c = (hi + lo)·W2 - c (x) wbar,  d' = W'·b2c = W'·b2 - c mean(b2)   (c = rowsum of the EXACT W')
            TP_TRY(pack_ln_fold_residual_launch(dt, inw + (size_t)(1 + g) * E * E * 2, gammas[g], (const float*)(P + L.b_kv2) + g * E, lo16,
                                                c_ex, d_ex, (int)E, (int)E, stream));
            GemmArgs a2 = plain_gemm(lo16, E, P + L.scratch_t, (char*)P2, E, (int)E, (int)E, (int)E, nullptr, 0);
            a2.tile = 128;
            TP_TRY(gemm_launch(TP_F16, TP_F32, a2, stream));
            // (g = 1, the V side: also as rows of interleaved K-tile pairs [hi_t | lo_t] (K = 2 E) for the absorbed schedule's per-head V GEMM)
            TP_TRY(pack_center_product_launch((const float*)(P + L.scratch_p), c_ex, wbar + g * (E + 1),
                                              P + L.w_cc_kv + (size_t)g * E * E * 2, d_ex,
                                              (float*)(P + L.d_cc_kv) + g * E, stream, sat, P2, g == 1 ? P + L.w_cc_v3 : nullptr));
        }
        TP_TRY(pack_head_transpose_launch(P + L.w_cc_kv, P + L.w_qt_cc, stream));
        // (the absorbed schedule on this chain: qt = per-head Q_h·Wc_k,h — the per-head transposes of the ROUNDED Wc_k)
        TP_TRY(pack_head_transpose_launch(P + L.w_c_kv, P + L.w_qt_c, stream));
        {   // query side: Q = rstd·(q0·Wcq^T − mu·c_q) + b'_q,  Wcq = W'q·Wq1  (q_proj_1 has no bias)
            TP_TRY(pack_transpose_f16_launch(P + L.w_q1, P + L.scratch_t, (int)E, stream));
            GemmArgs a = plain_gemm(P + L.w_in_q, E, P + L.scratch_t, P + L.scratch_p, E, (int)E, (int)E, (int)E, nullptr, 0);
            a.tile = 128;
            TP_TRY(gemm_launch(TP_F16, TP_F32, a, stream));
            TP_TRY(pack_round_f16_launch((const float*)(P + L.scratch_p), P + L.w_c_q, (long long)E * E, stream, sat));
            TP_TRY(pack_ln_fold_residual_launch(dt, inw, gammas[2], nullptr, lo16, c_ex, nullptr, (int)E, (int)E, stream));
            GemmArgs a2 = plain_gemm(lo16, E, P + L.scratch_t, (char*)P2, E, (int)E, (int)E, (int)E, nullptr, 0);
            a2.tile = 128;
            TP_TRY(gemm_launch(TP_F16, TP_F32, a2, stream));
            TP_TRY(pack_center_product_launch((const float*)(P + L.scratch_p), c_ex, wbar + 2 * (E + 1),
                                              P + L.w_cc_q, nullptr, nullptr, stream, sat, P2));
        }
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)(status + 2), 1, 1, stream);
        if (e != hipSuccess) { set_error("tp_pack_weights: hipMemsetD32Async(status): %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    // out_proj folded into mlp[0]:  W_om = Wm0·Wout (fp32 accumulate on the MFMA kernel, rounded once to fp16),
    // b_om = Wm0·bout + bm0 — in every inference pack (never in a training pack: a training step re-packs every step and
    // must not pay for a product it never uses); status[1] records it.
    if (!train_pack) {
        TP_TRY(pack_transpose_f16_launch(P + L.w_out, P + L.scratch_t, (int)E, stream));
        {
            GemmArgs a = plain_gemm(P + L.w_m0, E, P + L.scratch_t, P + L.scratch_p, E, D, (int)E, (int)E, nullptr, 0);
            a.tile = 128;
            TP_TRY(gemm_launch(TP_F16, TP_F32, a, stream));
        }
        TP_TRY(pack_round_f16_launch((const float*)(P + L.scratch_p), P + L.w_om, (long long)D * E, stream, sat));
        TP_TRY(pack_bias_fold_launch(P + L.w_m0, (const float*)(P + L.b_out), (const float*)(P + L.b_m0),
                                     (float*)(P + L.b_om), D, (int)E, stream));
        hipError_t e = hipMemsetD32Async((hipDeviceptr_t)(status + 1), 1, 1, stream);
        if (e != hipSuccess) { set_error("tp_pack_weights: hipMemsetD32Async(status): %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    pack_registry_put(packed, desc, train_pack);
    return TP_OK;
}

static int check_strides(const char* name, const void* p, const int64_t st[3]) {
    if (!p || !st) { set_error("%s: NULL pointer", name); return TP_ERR_INVALID_ARG; }
    if (st[2] != 1) { set_error("%s: innermost stride must be 1 (got %lld)", name, (long long)st[2]); return TP_ERR_INVALID_ARG; }
    if (st[1] % 8 != 0 || st[0] % 8 != 0 || ((uintptr_t)p & 15) != 0) {
        set_error("%s: base pointer must be 16-byte aligned and strides multiples of 8 elements", name);
        return TP_ERR_INVALID_ARG;
    }
    return TP_OK;
}

int tp_point_queries(const tp_desc* desc, const void* x, const int64_t x_strides[3], void* q0, void* stream) {
    TP_TRY(validate_desc(desc));
    TP_TRY(check_strides("x", x, x_strides));
    if (!q0) { set_error("tp_point_queries: q0 is NULL"); return TP_ERR_INVALID_ARG; }
    return point_queries_launch(desc->dtype, x, x_strides, q0, desc->batch, desc->raw_grid,
                                desc->scale_factor, (hipStream_t)stream);
}

int tp_region_attention(const tp_desc* desc, const void* q, const void* k, const void* v, void* o, void* stream) {
    TP_TRY(validate_desc(desc));
    if (!q || !k || !v || !o) { set_error("tp_region_attention: NULL pointer"); return TP_ERR_INVALID_ARG; }
    return region_attention_launch(q, k, v, o, desc->batch, desc->raw_grid, desc->scale_factor, (hipStream_t)stream);
}

int tp_region_attention_absorbed(const tp_desc* desc, const void* qt, const void* h2k, const void* h2v, const float* mr_k,
                                 const float* mr_v, void* u, void* stream) {
    TP_TRY(validate_desc(desc));
    if (!qt || !h2k || !h2v || !mr_k || !mr_v || !u) { set_error("tp_region_attention_absorbed: NULL pointer"); return TP_ERR_INVALID_ARG; }
    return region_attention_absorbed_launch(qt, h2k, h2v, mr_k, mr_v, u, desc->batch, desc->raw_grid, desc->scale_factor,
                                            (hipStream_t)stream);
}

int tp_hd_slice(const float* image, int H, int W, int h_block, int w_block, int h_res, int w_res, int hg, int wg,
                float* crops, int block, void* stream) {
    if (!image || !crops || H <= 0 || W <= 0 || h_block < 1 || w_block < 1 || block <= 0 || h_res <= 0 || w_res <= 0 ||
        h_res > h_block * block || w_res > w_block * block || (h_block * w_block > 1 && (hg <= 0 || wg <= 0 || hg > block || wg > block))) {
        set_error("tp_hd_slice: invalid argument");
        return TP_ERR_INVALID_ARG;
    }
    return hd_slice_launch(image, H, W, h_block, w_block, h_res, w_res, hg, wg, crops, block, (hipStream_t)stream);
}

int tp_debug_count_saturated(const tp_desc* desc, const void* workspace, size_t workspace_bytes, int32_t* counts, void* stream_) {
    TuningScope tuning_scope(desc);
    TP_TRY(validate_desc(desc));
    if (!workspace || !counts) { set_error("tp_debug_count_saturated: NULL argument"); return TP_ERR_INVALID_ARG; }
    const int B = desc->batch, g = desc->raw_grid, s = desc->scale_factor, D = desc->hidden_size;
    if (B > max_images_per_launch(desc)) { set_error("tp_debug_count_saturated: batch was served in chunks; scan a smaller batch"); return TP_ERR_INVALID_ARG; }
    const SchedulePlan plan = plan_schedule(desc, false, (desc->flags & TP_DESC_MASKED) != 0);    // (the schedule of the forward scanned)
    const WorkspaceLayout W = workspace_layout(B, g, s, D, plan);
    if (workspace_bytes < W.total) { set_error("tp_debug_count_saturated: workspace %zu B < %zu B", workspace_bytes, W.total); return TP_ERR_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    const long long rows_kv = (long long)B * g * g, rows_q = (long long)B * (g / s) * (g / s), E = kEmbed;
    const char* ws = (const char*)workspace;
    auto n_if = [](size_t off, long long n) -> long long { return off == kNoSlab ? 0 : n; };     // a slab the schedule does not have: nothing to scan
    const struct { size_t off; long long n; } bufs[TP_NUM_DEBUG_BUFFERS] = {
        {W.q0, rows_q * E}, {W.hkv, rows_kv * 2 * E},
        {W.h2, n_if(W.h2, 2 * rows_kv * E)},
        // (the absorbed schedule keeps qt | u [2][rows_q, 8, E] where K | V would be)
        {W.kv, n_if(W.kv, plan.absorb ? (plan.u_split ? 1 : 2) * rows_q * 8 * E : 2 * rows_kv * E)},      // (u_split: u alone; qt is fp32 there)
        {W.q1pre, n_if(W.q1pre, rows_q * E)},
        {W.q, rows_q * E}, {W.o, rows_q * E}, {W.a1, n_if(W.a1, rows_q * E)}, {W.a2, rows_q * (long long)D}};
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * TP_NUM_DEBUG_BUFFERS, stream);
    if (e != hipSuccess) { set_error("tp_debug_count_saturated: hipMemsetAsync: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    for (int i = 0; i < TP_NUM_DEBUG_BUFFERS; ++i)
        if (bufs[i].n > 0) TP_TRY(count_saturated_launch(ws + bufs[i].off, bufs[i].n, counts + i, stream));
    return TP_OK;
}

int64_t tp_hd_rows(int h_block, int w_block, int M) {
    if (h_block < 1 || w_block < 1 || M < 1) return 0;
    const int64_t n = (int64_t)h_block * w_block;
    return n * (M + 1) + (n > 1 ? M + 1 : 0);
}

int tp_hd_assemble(const tp_hd_image* plan, int n_images, const void* tokens, int64_t n_crops, const int32_t* crop_map,
                   const void* sep, const void* ret, void* out, int64_t out_rows, int M, int D, int dtype, void* stream) {
    if (!plan || !tokens || !sep || !ret || !out || n_images <= 0 || M <= 0 || n_crops <= 0 || out_rows <= 0) {
        set_error("tp_hd_assemble: NULL / non-positive argument");
        return TP_ERR_INVALID_ARG;
    }
    if (dtype != TP_BF16 && dtype != TP_F16) { set_error("tp_hd_assemble: dtype %d unsupported", dtype); return TP_ERR_INVALID_ARG; }
    if (D <= 0 || D % 8 != 0 || ((uintptr_t)tokens & 15) || ((uintptr_t)sep & 15) || ((uintptr_t)ret & 15) || ((uintptr_t)out & 15)) {
        set_error("tp_hd_assemble: D must be a multiple of 8 and every pointer 16-byte aligned");
        return TP_ERR_INVALID_ARG;
    }
    // Images come in order; their crop ranges and output row ranges must not overlap and must lie inside
    // tokens [n_crops, M, D] / out [out_rows, D].  GAPS between the output ranges are allowed and left untouched (the
    // text embeddings of inputs_embeds live there, llava_arch.py:172-191); gaps in the crop list are allowed too.
    int64_t crop = 0, row = 0;
    for (int i = 0; i < n_images; ++i) {
        if (plan[i].h_block < 1 || plan[i].w_block < 1 || plan[i].first_crop < crop || plan[i].out_row < row) {
            set_error("tp_hd_assemble: plan entry %d is inconsistent (overlaps its predecessor or is out of order)", i);
            return TP_ERR_INVALID_ARG;
        }
        crop = (int64_t)plan[i].first_crop + plan[i].h_block * plan[i].w_block + (plan[i].h_block * plan[i].w_block > 1 ? 1 : 0);
        row = plan[i].out_row + tp_hd_rows(plan[i].h_block, plan[i].w_block, M);
        if (crop > n_crops || row > out_rows) {
            set_error("tp_hd_assemble: plan entry %d reads crops up to %lld of %lld / writes rows up to %lld of %lld", i,
                      (long long)crop, (long long)n_crops, (long long)row, (long long)out_rows);
            return TP_ERR_INVALID_ARG;
        }
    }
    return hd_assemble_launch(plan, n_images, tokens, crop_map, sep, ret, out, M, D, (hipStream_t)stream);
}

int tp_ln_finalize(const float* row_stats, int parts, int64_t M, int ln_dim, float eps, float* row_mean_rstd, void* stream) {
    if (!row_stats || !row_mean_rstd || parts <= 0 || M <= 0 || ln_dim <= 0 || !(eps > 0.f)) {
        set_error("tp_ln_finalize: invalid argument");
        return TP_ERR_INVALID_ARG;
    }
    if (ln_dim != parts * 128) {
        set_error("tp_ln_finalize: ln_dim %d != %d slabs of 128 columns", ln_dim, parts);
        return TP_ERR_INVALID_ARG;
    }
    return ln_finalize_launch(row_stats, row_mean_rstd, M, parts, 1, ln_dim, eps, (hipStream_t)stream);
}

int tp_linear_stats_parts(const tp_linear_args* a) {
    if (!a || a->N <= 0 || a->N % 128 != 0) return 0;
    return gemm_stats_parts(a->N);
}

int tp_linear(const tp_linear_args* a, void* stream) {
    if (!a || !a->A || !a->W || (!a->C && !(a->flags & TP_LINEAR_NO_STORE))) { set_error("tp_linear: NULL argument"); return TP_ERR_INVALID_ARG; }
    if ((a->flags & TP_LINEAR_NO_STORE) && !(a->flags & TP_LINEAR_ROW_STATS)) {
        set_error("tp_linear: TP_LINEAR_NO_STORE only makes sense with TP_LINEAR_ROW_STATS");
        return TP_ERR_INVALID_ARG;
    }
    if ((a->flags & TP_LINEAR_LN_FOLD) && (!a->row_mean_rstd || !a->colsum)) {
        set_error("tp_linear: LN_FOLD needs row_mean_rstd and colsum");
        return TP_ERR_INVALID_ARG;
    }
    if ((a->flags & TP_LINEAR_ROW_STATS) && !a->row_stats_out) {
        set_error("tp_linear: ROW_STATS needs row_stats_out");
        return TP_ERR_INVALID_ARG;
    }
    if (a->lda % 8 != 0 || a->a_batch_stride % 8 != 0 || ((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15) ||
        ((uintptr_t)a->C & 15) || a->ldc % 8 != 0) {
        set_error("tp_linear: A/W/C must be 16-byte aligned, lda, ldc and batch stride multiples of 8 elements");
        return TP_ERR_INVALID_ARG;
    }
    GemmArgs g{};
    g.A = (const char*)a->A; g.W = (const char*)a->W; g.C = (char*)a->C;
    g.bias = a->bias; g.stats_in = a->row_mean_rstd; g.colsum = a->colsum; g.stats_out = a->row_stats_out;
    g.rows_per_batch = (a->rows_per_batch > 0 && a->rows_per_batch < a->M) ? a->rows_per_batch : a->M;
    g.a_batch_stride_bytes = a->a_batch_stride * 2; g.lda_bytes = a->lda * 2; g.ldc = a->ldc;
    if (a->ldw != 0) {
        if (a->ldw < a->K || a->ldw % 8 != 0) { set_error("tp_linear: ldw must be 0 or a multiple of 8 elements >= K"); return TP_ERR_INVALID_ARG; }
        g.ldw_bytes = (long long)a->ldw * 2;
    }
    g.M = a->M; g.N = a->N; g.K = a->K; g.flags = a->flags;
    g.groups = 1; g.tile = a->tile;
    if (a->a_k_dup != 0) {
        if (a->a_k_dup < 0 || a->a_k_dup % 64 != 0 || 2 * (long long)a->a_k_dup > a->K) { set_error("tp_linear: a_k_dup must be a multiple of 64 and at most K / 2"); return TP_ERR_INVALID_ARG; }
        g.a_k_dup = a->a_k_dup;
    }
    if (a->flags & TP_LINEAR_OUT_F32) { set_error("tp_linear: TP_LINEAR_OUT_F32 was replaced by out_dtype = TP_F32"); return TP_ERR_INVALID_ARG; }
    return gemm_launch(a->dtype, a->out_dtype, g, (hipStream_t)stream);
}

}  // extern "C"  (forward_impl has C++ linkage: tp_train.hip calls it too)

namespace tp {

// The query side of the path (point queries -> q_proj_1 -> LayerNorm -> q in-projection, ~8 % of a forward) does
// not depend on the K/V side until the attention kernel.  It is enqueued on a side stream forked from / joined
// back into the caller's stream with two events, so its small launches fill the tails of the K/V side's
// persistent GEMMs instead of running alone at 2.25 CU rounds.  One side stream + event pair per caller stream,
// created on first use (the only state the library keeps besides the tuning table and the error string).
struct SideCtx { hipStream_t s; hipEvent_t fork, join, kv0; };    // (kv0: the first K/V layer is done — the decoupled K launch's cue)
// `pins`: forwards between their fork and the enqueue of their join on this entry; `doomed`: released while pinned — destroyed by
// the last unpin.  A pinned entry is never evicted and never destroyed under a forward that still records / waits on its events.
struct SideEntry { int dev; hipStream_t main; SideCtx ctx; unsigned long long stamp; int pins; bool doomed; };
static std::mutex g_side_mu;
static std::vector<SideEntry> g_side_cache;
static unsigned long long g_side_clock = 0;
static void side_ctx_destroy(const SideCtx& c) {    // OUTSIDE the lock: the synchronize may block, and must not stall other forwards' lookups
    (void)hipStreamSynchronize(c.s);                // whatever was forked onto it has been joined by its forward; be sure anyway
    (void)hipEventDestroy(c.fork);
    (void)hipEventDestroy(c.join);
    (void)hipEventDestroy(c.kv0);
    (void)hipStreamDestroy(c.s);
}
// The side stream of (device, caller stream), created on first use and PINNED until side_unpin(); at most 64 per process, the
// least recently used UNPINNED one first out (none unpinned: this forward runs its query side on the caller's stream).
static bool side_ctx_for(hipStream_t main, SideCtx* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    SideCtx victim{};
    bool have_victim = false, ok = false;
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        for (auto& e : g_side_cache)
            if (e.dev == dev && e.main == main && !e.doomed) { e.stamp = ++g_side_clock; ++e.pins; *out = e.ctx; return true; }
        bool room = g_side_cache.size() < 64;
        if (!room) {
            long lru = -1;
            for (size_t i = 0; i < g_side_cache.size(); ++i)
                if (g_side_cache[i].pins == 0 && (lru < 0 || g_side_cache[i].stamp < g_side_cache[(size_t)lru].stamp)) lru = (long)i;
            if (lru >= 0) {
                victim = g_side_cache[(size_t)lru].ctx; have_victim = true;
                g_side_cache.erase(g_side_cache.begin() + lru);
                room = true;
            }
        }
        if (room) {
            SideCtx c{};
            if (hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) == hipSuccess) {
                if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess) (void)hipStreamDestroy(c.s);
                else if (hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(c.fork); (void)hipStreamDestroy(c.s); }
                else if (hipEventCreateWithFlags(&c.kv0, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(c.fork); (void)hipEventDestroy(c.join); (void)hipStreamDestroy(c.s); }
                else { g_side_cache.push_back(SideEntry{dev, main, c, ++g_side_clock, 1, false}); *out = c; ok = true; }
            }
        }
    }
    if (have_victim) side_ctx_destroy(victim);
    return ok;
}
static void side_unpin(hipStream_t main, const SideCtx& ctx) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    SideCtx victim{};
    bool have_victim = false;
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        for (size_t i = 0; i < g_side_cache.size(); ++i) {
            SideEntry& e = g_side_cache[i];
            if (e.dev == dev && e.main == main && e.ctx.s == ctx.s) {
                if (e.pins > 0) --e.pins;
                if (e.doomed && e.pins == 0) { victim = e.ctx; have_victim = true; g_side_cache.erase(g_side_cache.begin() + (long)i); }
                break;
            }
        }
    }
    if (have_victim) side_ctx_destroy(victim);
}
struct SidePin {                                    // unpins when the forward leaves (its join has been enqueued, or it failed)
    hipStream_t main; SideCtx ctx; bool held;
    ~SidePin() { if (held) side_unpin(main, ctx); }
};
int release_stream_state(hipStream_t main) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("tp_release_stream: hipGetDevice failed"); return TP_ERR_LAUNCH; }
    SideCtx victim{};
    bool have_victim = false;
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        for (size_t i = 0; i < g_side_cache.size(); ++i)
            if (g_side_cache[i].dev == dev && g_side_cache[i].main == main && !g_side_cache[i].doomed) {
                if (g_side_cache[i].pins > 0) g_side_cache[i].doomed = true;       // a forward of another thread is inside: its unpin destroys
                else { victim = g_side_cache[i].ctx; have_victim = true; g_side_cache.erase(g_side_cache.begin() + (long)i); }
                break;
            }
    }
    if (have_victim) side_ctx_destroy(victim);
    return TP_OK;
}
int side_cache_size() { std::lock_guard<std::mutex> lock(g_side_mu); return (int)g_side_cache.size(); }

int forward_impl(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
                 const int64_t xm_strides[3], const void* packed_weights, void* out, void* workspace,
                 size_t workspace_bytes, void* stream_, void* const* stage_events, bool train,
                 const void* const* xm_parts, const float* attn_mask, int mask_mode) {
    TuningScope tuning_scope(desc);
    TP_TRY(validate_desc(desc));
    TP_TRY(check_strides("x", x, x_strides));
    if (xm_parts) {                                     // x_multi given as the four [B, N, 1024] hidden-state slices
        for (int i = 0; i < 4; ++i) TP_TRY(check_strides("x_multi part", xm_parts[i], xm_strides));
        x_multi = xm_parts[0];
    } else {
        TP_TRY(check_strides("x_multi", x_multi, xm_strides));
    }
    if (!packed_weights || !out || !workspace) { set_error("tp_forward: NULL argument"); return TP_ERR_INVALID_ARG; }
    {
        // A GEMM launch addresses at most 4 GiB of output (range-checked 32-bit store offsets).  No operation mixes
        // images, so a batch beyond that is served as consecutive chunks of the same call — bit-identical per image.
        const long long n_tok = (long long)desc->raw_grid * desc->raw_grid, m_tok = n_tok / ((long long)desc->scale_factor * desc->scale_factor);
        const long long out_esz = desc->out_dtype == TP_F32 ? 4 : 2;
        const long long bc = max_images_per_launch(desc);
        if (bc < 1) { set_error("tp_forward: a single image exceeds the 4 GiB a GEMM launch can address"); return TP_ERR_INVALID_ARG; }
        if (desc->batch > bc) {
            if (mask_mode == 2) { set_error("tp_forward: a per-head attn_mask needs the batch in one launch (%d > %lld images)", desc->batch, bc); return TP_ERR_INVALID_ARG; }
            if (train || stage_events) {
                set_error("tp_forward: batch %d exceeds %lld images per call for the training / staged entry points", desc->batch, bc);
                return TP_ERR_INVALID_ARG;
            }
            const long long x_esz = 2;
            for (long long b0 = 0; b0 < desc->batch; b0 += bc) {
                tp_desc d = *desc;
                d.batch = (int)((desc->batch - b0) < bc ? (desc->batch - b0) : bc);
                const void* parts[4];
                if (xm_parts) for (int i = 0; i < 4; ++i) parts[i] = (const char*)xm_parts[i] + b0 * xm_strides[0] * x_esz;
                TP_TRY(forward_impl(&d, (const char*)x + b0 * x_strides[0] * x_esz, x_strides,
                                    xm_parts ? nullptr : (const char*)x_multi + b0 * xm_strides[0] * x_esz, xm_strides, packed_weights,
                                    (char*)out + b0 * m_tok * desc->hidden_size * out_esz, workspace, workspace_bytes, stream_,
                                    nullptr, false, xm_parts ? parts : nullptr,
                                    // (a per-(region, image, head) mask is laid out for the WHOLE batch: it cannot be chunked)
                                    attn_mask, mask_mode));
            }
            return TP_OK;
        }
    }
    const int B = desc->batch, g = desc->raw_grid, s = desc->scale_factor, D = desc->hidden_size, dt = desc->dtype;
    const int N = g * g, G = g / s, M = G * G, E = kEmbed;
    const int rows_kv = B * N, rows_q = B * M;
    const PackedLayout P = packed_layout(D);
    tp_desc dplan = *desc; dplan.batch = B;
    const SchedulePlan plan = plan_schedule(&dplan, train, attn_mask != nullptr);
    const WorkspaceLayout W = workspace_layout(B, g, s, D, plan);
    if (workspace_bytes < W.total) {
        set_error("tp_forward: workspace %zu B < required %zu B%s", workspace_bytes, W.total,
                  attn_mask ? " (a masked forward: size it with TP_DESC_MASKED in tp_desc.flags)" : "");
        return TP_ERR_WORKSPACE;
    }
    TP_TRY(pack_registry_check(packed_weights, desc, train));
    if (((uintptr_t)workspace & 255) || ((uintptr_t)packed_weights & 255) || ((uintptr_t)out & 15)) {
        set_error("tp_forward: workspace/packed buffers must be 256-byte aligned, out 16-byte aligned");
        return TP_ERR_INVALID_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const char* pw = (const char*)packed_weights;
    char* ws = (char*)workspace;
    auto slab = [&](size_t off) -> char* { return off == kNoSlab ? nullptr : ws + off; };      // a slab the schedule does not have: NULL
    const long long kvE = (long long)rows_kv * E;      // elements per K/V group slab
    // Hkv: training keeps [rows, 2048] (K half | V half per row: what the backward reads); inference writes a K slab and a V
    // slab [rows, 1024] each (GemmArgs::c_split_cols), so that everything behind the first layer walks contiguous rows
    const bool hkv_split = !train;
    const long long hkv_ld = hkv_split ? E : 2 * E;                     // elements between rows of a half
    const long long hkv_gs = hkv_split ? kvE * 2 : (long long)E * 2;    // bytes from the K half to the V half

    // tile-queue heads of the persistent GEMM launches: 64 ints per launch (<= 16 launches), zeroed once per forward
    // A persistent launch reads its queue head only when it has more tiles than workgroups (tp_gemm8.hip: "one tile per workgroup"
    // otherwise).  A batch so small that NO launch of the forward can (counted in 128-row half tiles, the finest the routes use, on
    // the widest weight of either side) needs neither the heads nor the fill kernel in front of them — ~10 us of a 0.13 ms forward.
    bool queues_possible = true;
    if (!train) {
        const long long nwg = gemm8_persistent_cus();
        const long long cols_q = std::max<long long>(D / 256, plan.absorb ? 8 * E / 256 : E / 256);
        queues_possible = (rows_kv + 127) / 128 * (2 * E / 256) > nwg || (rows_q + 127) / 128 * cols_q > nwg;
    }
    int* counters = (tuning(TP_TUNE_DYNAMIC_TILES) && queues_possible) ? (int*)(ws + W.counters) : nullptr;
    if (counters) {
        hipError_t e = hipMemsetAsync(counters, 0, kCounterBytes, stream);
        if (e != hipSuccess) { set_error("tp_forward: hipMemsetAsync: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    int launch_no = 0;
    int stage_idx = 0;                                  // advanced by mark(): 1 .. TP_NUM_STAGES while stage k - 1 is being enqueued
    auto launch = [&](int in_dt, int out_dt, GemmArgs& a, hipStream_t st) -> int {
        if (counters && launch_no < kMaxLaunches) {
            a.tile_counters = counters + 64 * launch_no;                         // [groups <= 8][8 XCDs] heads per launch
            ++launch_no;
        } else a.tile_counters = nullptr;
        a.sat_flag = (int*)(ws + W.status); a.sat_bit = 1 << stage_idx;          // sticky fp16-saturation bits, by stage (bit 0: query side)
        return gemm_launch(in_dt, out_dt, a, st);
    };
    // TP_TUNE_SPLIT_K: a latency-bound K = 4096 GEMM of a small batch as S K-groups of the 128-tile kernel (fp32 partials)
    // + one reduction kernel that applies the epilogue.  `a` is the un-split launch (bias, GELU flag, destination).
    auto launch_maybe_splitk = [&](int in_dt, int out_dt, GemmArgs& a, hipStream_t st) -> int {
        const long long tiles = (long long)((a.M + 127) / 128) * (a.N / 128);
        int S = 1;
        if (plan.split_k && a.groups == 1 && (a.flags & ~TP_LINEAR_GELU) == 0)
            while (S < 8 && tiles * (S * 2) <= 512 && a.K / (S * 2) >= 4 * BK_ELEMS && a.K % (S * 2 * BK_ELEMS) == 0) S *= 2;
        if (S < 4) return launch(in_dt, out_dt, a, st);    // (two K-groups do not pay for the partials' round trip: measured)
        GemmArgs p = a;
        p.groups = S; p.K = a.K / S;
        p.a_gs = a.A_parts[0] ? 0 : (long long)p.K * 2; p.w_gs = (long long)p.K * 2;      // (four-part A: the kernel walks the sources)
        p.parts_k_groups = a.A_parts[0] ? 1 : 0;
        p.ldw_bytes = a.ldw_bytes ? a.ldw_bytes : (long long)a.K * 2;
        p.C = slab(W.splitk); p.ldc = a.N; p.c_gs = (long long)a.M * a.N * 4;
        p.bias = nullptr; p.flags = 0; p.tile = 128; p.c_split_cols = 0;
        TP_TRY(launch(in_dt, TP_F32, p, st));
        return splitk_reduce_launch((const float*)slab(W.splitk), S, a.M, a.N, a.bias, (a.flags & TP_LINEAR_GELU) ? 1 : 0, a.C, a.ldc,
                                    out_dt, st, a.c_split_cols, a.c_split_stride_bytes / (out_dt == TP_F32 ? 4 : 2),
                                    (int*)(ws + W.status), 1 << stage_idx);
    };
    // query side on a side stream (not when the caller wants per-stage events: those need one stream)
    SideCtx side_storage{};
    SideCtx* side = (tuning(TP_TUNE_Q_SIDE_STREAM) && !stage_events && side_ctx_for(stream, &side_storage)) ? &side_storage : nullptr;
    SidePin side_pin{stream, side_storage, side != nullptr};   // pinned until this call returns: nobody destroys its events under it
    const int parts_q = gemm_stats_parts(E);
    // (fused LayerNorm chain, inference: Q1pre is computed for its row statistics only; the in-projection reads q0)
    const bool fuse_q = plan.fuse_q;
    const bool tri = plan.tri;                          // centred chain weights + triangular statistics GEMMs (tp_pack_qr.hip): no mean anywhere
    auto q_proj = [&](hipStream_t st) -> int {          // 5. Q1pre = q0 · Wq1^T (no bias), LayerNorm partials
        // (triangular statistics: the weight is R of W2c = Q R — the row's variance is the second moment of q0·R^T)
        GemmArgs a = plain_gemm(ws + W.q0, E, tri ? pw + P.w_r_q : pw + P.w_q1, slab(W.q1pre), E, rows_q, E, E, nullptr,
                                TP_LINEAR_ROW_STATS | (fuse_q ? TP_LINEAR_NO_STORE : 0));
        a.tri = tri ? 1 : 0;
        a.stats_out = (float*)(ws + W.stats_q);
        return launch(TP_F16, TP_F16, a, st);
    };
    // Small M (the 128-tile kernel): the consumer of a LayerNorm merges the producer's (mean, M2) slabs itself — one launch
    // less per LayerNorm on a path where, at B = 1, every launch is 6 % of the forward.  Inference only: the backward reads
    // the (mean, rstd) buffers.
    auto merge_in_kernel = [&](GemmArgs& a, const float* slabs, long long slabs_gs) -> bool {
        if (train || tuning(TP_TUNE_LN_MERGE) == 1 || !gemm_uses_small_kernel(a)) return false;
        a.stats_parts = slabs; a.stats_parts_gs = slabs_gs; a.ln_inv_dim = 1.0f / E; a.ln_eps = desc->ln_eps;
        a.ln_second_moment = tri ? 1 : 0;
        a.stats_in = nullptr;
        return true;
    };
    static_assert(kEmbed / 128 == 8, "ln_merge_slabs<8>");
    auto q_inproj = [&](hipStream_t st) -> int {        // 6. Q = LN(Q1pre) · Winq^T + b
        GemmArgs a = plain_gemm(fuse_q ? ws + W.q0 : slab(W.q1pre), E, fuse_q ? (tri ? pw + P.w_cc_q : pw + P.w_c_q) : pw + P.w_in_q, ws + W.q, E,
                                rows_q, E, E, (const float*)(pw + P.b_in_q), TP_LINEAR_LN_FOLD);
        a.stats_in = (const float*)(ws + W.mr_q);
        a.colsum = (const float*)(pw + P.c_in_q);
        if (!merge_in_kernel(a, (const float*)(ws + W.stats_q), 0))
            TP_TRY(ln_finalize_launch((const float*)(ws + W.stats_q), (float*)(ws + W.mr_q), rows_q, parts_q, 1, E,
                                      desc->ln_eps, st, tri));
        return launch(TP_F16, TP_F16, a, st);
    };
    const bool absorb = plan.absorb;
    // (fused LayerNorm chain — inference, plain schedule: H2 is needed for its row statistics only and is not written)
    const bool fuse_ln = plan.fuse_ln;
    // scale_factor 2 on that chain: region-major K/V rows, and region attention inside the in-projections' epilogues
    // (the attention epilogues want 8 | rows per image — region pairs in the query DMA —: every even grid but 6, 10, 14, …;
    // decided per image, not per batch, so that an image's bits do not depend on the batch it travels in)
    // (absorbed schedule on the fused chain: H2 for its statistics only, the attention kernel walks Hkv — RAW form)
    const bool absorb_raw = plan.absorb_raw;
    const bool region_major = plan.region_major;
    const bool fuse_attn = plan.fuse_attn;
    char* const kv_slab = slab(W.kv);                                  // K | V, or qt | u on the absorbed schedule (NULL: neither)
    // absorbed schedule: qt [rows_q, 8, E] fp16 | u [rows_q, 8, E] fp16;  u_split (the default of s >= 3): u [rows_q, 8, E] as ONE fp16
    // value per element, followed by qt [rows_q, 8, E] in FP32 — a 3 x [rows_q, 8, E] x 2 B slab (round 4: the 128-seed parity sweep
    // has its tail on the logit side, where Q and qt were rounded one behind the other; round 5 dropped u's hi | lo halves)
    const bool u_split = plan.u_split;
    char* const uu = !kv_slab ? nullptr : (u_split ? kv_slab : kv_slab + (size_t)rows_q * 8 * E * 2);
    char* const qt = !kv_slab ? nullptr : (u_split ? kv_slab + (size_t)rows_q * 8 * E * 2 : kv_slab);
    auto qt_gemm = [&](hipStream_t st) -> int {         // qt[m, h, :] = Q[m, h*128:(h+1)*128] · W'k[h*128:(h+1)*128, :]
        GemmArgs a = plain_gemm(ws + W.q, E, absorb_raw ? (tri ? pw + P.w_qt_cc : pw + P.w_qt_c) : pw + P.w_qt, qt, 8 * E, rows_q, E, kHeadDim, nullptr, 0);
        a.groups = kHeads; a.a_gs = kHeadDim * 2; a.w_gs = (long long)E * kHeadDim * 2; a.c_gs = E * (u_split ? 4 : 2);
        return launch(TP_F16, u_split ? TP_F32 : TP_F16, a, st);
    };
    // decoupled K launch (TP_TUNE_DECOUPLE_K = 1 / 2, GemmArgs::attn_decoupled; round 6): it needs Hkv and Q but not
    // the K/V row statistics, so with 1 it follows the query side on the side stream — beside the statistics launch of the caller's
    // stream instead of behind it; 2 keeps it on the caller's stream (same bits; the A/B of the placement alone).  MEASURED NULL
    // (profiles/r06b_decouple_k_ab.txt, r06c_timeline_B32.txt): at a 32-image shard the statistics launch, the query-side GEMMs and
    // the K launch are each full-chip work (the router already removes their tail rounds with half tiles), so side-by-side placement
    // only trades the K launch's wait for a 13 us cross-stream join; 0.665 (1) / 0.658 (2) / 0.661 ms (0).  Off by default.
    const bool decouple_k = plan.decouple_k;
    const bool k_on_side = decouple_k && side != nullptr && tuning(TP_TUNE_DECOUPLE_K) == 1;
    // 2. Hkv = GELU(x_multi · [Wk0;Wv0]^T + b): strided A (tower hands over [:,1:] slices)
    auto kv_layer0 = [&]() -> int {
        GemmArgs a = plain_gemm(x_multi, xm_strides[1], pw + P.w_kv0, ws + W.hkv, 2 * E, rows_kv, 2 * E, kMulti,
                                (const float*)(pw + P.b_kv0), TP_LINEAR_GELU | (train ? TP_LINEAR_SAVE_PRE : 0));
        a.rows_per_batch = N; a.a_batch_stride_bytes = xm_strides[0] * 2;
        if (region_major) { a.a_region_g = g; a.a_region_s = s; }      // rows of Hkv (and of everything behind it) by region
        if (hkv_split) { a.ldc = E; a.c_split_cols = E; a.c_split_stride_bytes = kvE * 2; }
        a.C2 = train ? ws + W.z1 : nullptr;
        if (xm_parts) {
            for (int i = 0; i < 4; ++i) a.A_parts[i] = (const char*)xm_parts[i];
            a.k_part = kMulti / 4;
        }
        const int keep = stage_idx;
        stage_idx = 2;                                   // (its saturation bit is stage kv_layer0's, wherever in the enqueue order it goes out)
        const int rc = launch_maybe_splitk(dt, TP_F16, a, stream);       // raw operands in the io dtype, fp16 activations out
        stage_idx = keep;
        return rc;
    };
    if (side) {
        // The fork event is recorded FIRST, then the first K/V layer goes out on the caller's stream, then the side stream's launches
        // (round 6): a one-image forward is bound by the HOST's enqueue rate (11 launches of 5-25 us each), and with the query side's
        // three launches enqueued in front of it the first layer — the head of the critical path — started 23 us late
        // (profiles/r06p_timeline_B1.txt).  Measured +-0 (0.1277 vs 0.1282 ms at B = 1): the host enqueues just in time either way;
        // kept because it is the right order.  The side stream still waits only for what preceded this forward.
        hipError_t e = hipEventRecord(side->fork, stream);
        if (e != hipSuccess) { set_error("tp_forward: side stream fork: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
        TP_TRY(kv_layer0());
        e = hipStreamWaitEvent(side->s, side->fork, 0);
        if (e != hipSuccess) { set_error("tp_forward: side stream fork: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
        TP_TRY(point_queries_launch(dt, x, x_strides, ws + W.q0, B, g, s, side->s));
        TP_TRY(q_proj(side->s));
        TP_TRY(q_inproj(side->s));
        if (absorb) TP_TRY(qt_gemm(side->s));
        if (!k_on_side) {
            e = hipEventRecord(side->join, side->s);
            if (e != hipSuccess) { set_error("tp_forward: side stream join: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
        }
    }
    auto mark = [&]() -> int {
        if (stage_events) {
            hipError_t e = hipEventRecord((hipEvent_t)stage_events[stage_idx], stream);
            if (e != hipSuccess) { set_error("hipEventRecord(stage %d): %s", stage_idx, hipGetErrorString(e)); return TP_ERR_LAUNCH; }
        }
        ++stage_idx;
        return TP_OK;
    };
    TP_TRY(mark());
    // 1. coarse point queries
    if (!side) TP_TRY(point_queries_launch(dt, x, x_strides, ws + W.q0, B, g, s, stream));
    // (attention in the in-projections' epilogues: the K launch of step 4 needs Q — on one stream the whole query side runs here)
    if (!side && fuse_attn) { TP_TRY(q_proj(stream)); TP_TRY(q_inproj(stream)); }

    TP_TRY(mark());
    // 2. (kv_layer0: enqueued above, in front of the side stream's launches, when there is a side stream)
    if (!side) TP_TRY(kv_layer0());
    if (k_on_side) {                                    // the decoupled K launch's cue (enqueued further down, on the side stream)
        hipError_t e = hipEventRecord(side->kv0, stream);
        if (e != hipSuccess) { set_error("tp_forward: hipEventRecord(kv0): %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    // 3. H2[g] = Hkv[:, g*1024:(g+1)*1024] · W{k,v}2^T + b, and LayerNorm partials of H2
    TP_TRY(mark());
    const int parts_kv = gemm_stats_parts(E);
    {
        const bool stats_only = fuse_ln || absorb_raw;
        const bool tri_kv = tri && stats_only;
        GemmArgs a = plain_gemm(ws + W.hkv, hkv_ld, tri_kv ? pw + P.w_r_kv : pw + P.w_kv2, stats_only ? nullptr : slab(W.h2), E, rows_kv, E, E,
                                tri_kv ? (const float*)(pw + P.c_r_kv) : (const float*)(pw + P.b_kv2),
                                TP_LINEAR_ROW_STATS | (stats_only ? TP_LINEAR_NO_STORE : 0));
        a.tri = tri_kv ? 1 : 0;
        a.groups = 2; a.a_gs = hkv_gs; a.w_gs = (long long)E * E * 2; a.c_gs = kvE * 2; a.bias_gs = E;
        a.stats_out = (float*)(ws + W.stats_kv); a.stats_out_gs = (long long)parts_kv * rows_kv * 2;
        TP_TRY(launch(TP_F16, TP_F16, a, stream));
    }
    TP_TRY(mark());
    bool kv_finalized = false;
    auto kv_finalize = [&]() -> int {                   // (mean, rstd) of both K/V groups, unless the consumer merges the slabs
        if (kv_finalized) return TP_OK;
        kv_finalized = true;
        return ln_finalize_launch((const float*)(ws + W.stats_kv), (float*)(ws + W.mr_kv), rows_kv, parts_kv, 2, E,
                                  desc->ln_eps, stream, tri && (fuse_ln || absorb_raw), decouple_k);
    };
    bool joined = false;
    auto join_side = [&]() -> int {
        if (side && !joined) {
            hipError_t e = hipStreamWaitEvent(stream, side->join, 0);
            if (e != hipSuccess) { set_error("tp_forward: side stream join wait: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
        }
        joined = true;
        return TP_OK;
    };
    // 4. {K,V} = LN(H2[g]) · Win{k,v}^T + b   (LayerNorm folded into the epilogue) — not in the absorbed schedule
    auto attn_gemm = [&](const int kv, hipStream_t st) -> int {
        // steps 4 + 7 as two launches: the K launch (step 4) turns its tile of K into logits against the region's query, the
        // V launch (step 7) its tile of V into softmax-weighted sums -> O.  K and V are never written.
        GemmArgs a = plain_gemm(ws + W.hkv + (size_t)kv * hkv_gs, hkv_ld, (tri ? pw + P.w_cc_kv : pw + P.w_c_kv) + (size_t)kv * E * E * 2,
                                kv == 0 ? nullptr : ws + W.o, E, rows_kv, E, E, (const float*)(pw + P.b_in_kv) + kv * E,
                                TP_LINEAR_LN_FOLD);
        a.acc_init = (const float*)(tri ? pw + P.d_cc_kv : pw + P.d_in_kv) + kv * E;
        a.stats_in = (const float*)(ws + W.mr_kv) + (size_t)kv * rows_kv * 2;
        a.colsum = (const float*)(pw + P.c_in_kv) + kv * E;
        a.attn_mode = kv + 1;
        a.attn_q = ws + W.q; a.attn_ldq_bytes = E * 2;
        a.attn_logits = (float*)(ws + W.attn_aux);      // [8 heads][rows_kv] fp32
        a.attn_scale = 0.08838834764831845f;            // 1/sqrt(128): q scaling of F.multi_head_attention_forward
        a.attn_decoupled = decouple_k ? 1 : 0;
        if (decouple_k && kv == 0) return launch(TP_F16, TP_F16, a, st);     // (reads no statistics: stats_in is staged, never used)
        if (!merge_in_kernel(a, (const float*)(ws + W.stats_kv) + (size_t)kv * parts_kv * rows_kv * 2, 0)) TP_TRY(kv_finalize());
        else if (decouple_k) a.attn_kstats_parts = (const float*)(ws + W.stats_kv);      // (V launch: the K group's slabs as well)
        return launch(TP_F16, TP_F16, a, st);
    };
    if (fuse_attn && !k_on_side) {
        TP_TRY(join_side());                            // Q must be there
        TP_TRY(attn_gemm(0, stream));
    } else if (fuse_attn) {                             // decoupled: behind the query side on the side stream, beside the statistics launch
        hipError_t e = hipStreamWaitEvent(side->s, side->kv0, 0);
        if (e != hipSuccess) { set_error("tp_forward: side stream wait(kv0): %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
        TP_TRY(attn_gemm(0, side->s));
        e = hipEventRecord(side->join, side->s);
        if (e != hipSuccess) { set_error("tp_forward: side stream join: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    } else
    if (!absorb) {
        TP_TRY(kv_finalize());
        GemmArgs a = plain_gemm(slab(W.h2), E, pw + P.w_in_kv, kv_slab, E, rows_kv, E, E,
                                (const float*)(pw + P.b_in_kv), TP_LINEAR_LN_FOLD);
        a.groups = 2; a.a_gs = kvE * 2; a.w_gs = (long long)E * E * 2; a.c_gs = kvE * 2; a.bias_gs = E;
        if (fuse_ln) {                                  // {K,V} = rstd·(Hkv[:, g]·Wc^T + d − mu·c) + b'
            a.A = (const char*)(ws + W.hkv); a.lda_bytes = hkv_ld * 2; a.a_gs = hkv_gs;
            a.W = tri ? pw + P.w_cc_kv : pw + P.w_c_kv;
            a.acc_init = (const float*)(tri ? pw + P.d_cc_kv : pw + P.d_in_kv); a.acc_init_gs = E;
        }
        a.stats_in = (const float*)(ws + W.mr_kv); a.stats_in_gs = (long long)rows_kv * 2;
        a.colsum = (const float*)(pw + P.c_in_kv); a.colsum_gs = E;
        TP_TRY(launch(TP_F16, TP_F16, a, stream));
    }
    TP_TRY(mark());
    if (!side && !fuse_attn) TP_TRY(q_proj(stream));
    TP_TRY(mark());
    if (!side && !fuse_attn) { TP_TRY(q_inproj(stream)); if (absorb) TP_TRY(qt_gemm(stream)); }
    TP_TRY(mark());
    TP_TRY(join_side());
    // 7. region-to-point attention
    if (fuse_attn) {
        TP_TRY(attn_gemm(1, stream));
    } else if (absorb) {
        TP_TRY(kv_finalize());
        float* const mr_u = (float*)(ws + W.attn_aux);     // RAW: (e / a, a) per head and query
        if (absorb_raw)
            TP_TRY(region_attention_absorbed_launch(qt, ws + W.hkv, ws + W.hkv + (size_t)hkv_gs, (const float*)(ws + W.mr_kv),
                                                    (const float*)(ws + W.mr_kv) + (size_t)rows_kv * 2, uu, B, g, s, stream, attn_mask, mask_mode,
                                                    (int)hkv_ld, ws + W.q, (const float*)(tri ? pw + P.d_cc_kv : pw + P.d_in_kv),
                                                    (const float*)(pw + P.c_in_kv), mr_u, u_split));
        else
            TP_TRY(region_attention_absorbed_launch(qt, slab(W.h2), slab(W.h2) + kvE * 2, (const float*)(ws + W.mr_kv),
                                                    (const float*)(ws + W.mr_kv) + (size_t)rows_kv * 2, uu, B, g, s, stream, attn_mask, mask_mode));
        // O[:, h*128:(h+1)*128] = u[:, h, :] · W'v[h*128:(h+1)*128, :]^T + b'v   (eight N = 128 groups)
        // RAW: a_h (u_h · Wc_v,h^T + d_v,h - (e_h / a_h) c_v,h) + b'v,h — a LayerNorm-fold epilogue with (mean, rstd) := mr_u
        // u_split (round 5 form): u is ONE fp16 value per element, the pre-multiplied weight stays hi + lo — W's rows are the K-tile
        // pairs [hi_0 lo_0 hi_1 lo_1 .. hi_15 lo_15] (K = 2 E) and GemmArgs::a_k_dup = E lets every K-tile of u serve its pair back to
        // back (the second fetch hits the L2): u_hi·W_hi + u_hi·W_lo.  Round 4 also carried u's own rounding residual (hi | lo halves,
        // K = 3 E): over 128 seeds x {s = 3, 4} x {bf16, fp16} the worst seed is 8.4e-4 .. 9.2e-4 without it against 8.1e-4 .. 9.2e-4 with
        // it (medians +5 %: 6.2-6.7e-4; dropping the WEIGHT's residual instead, or both, puts the worst seed at 9.3e-4 .. 9.6e-4) — and
        // it cost 268 MB written by the attention kernel + 268 MB read here per B = 256, s = 3 forward in two HBM-bound kernels:
        // attention stage 0.52 -> 0.40 ms (profiles/r05r_u_mode_ab.json).  Large launches take the pair kernel (its 256 x 128 tile fits
        // N = 128 per head), small ones the 128-tile kernel — one K order, the same bits (gemm_takes_pair_route).
        const int uK = u_split ? 2 * E : E, uld = E;
        GemmArgs a = plain_gemm(uu, 8 * uld, u_split ? pw + P.w_cc_v3
                                                     : (absorb_raw ? (tri ? pw + P.w_cc_kv : pw + P.w_c_kv) : pw + P.w_in_kv) + (size_t)E * E * 2,
                                ws + W.o, E, rows_q, kHeadDim, uK, (const float*)(pw + P.b_in_kv) + E, absorb_raw ? TP_LINEAR_LN_FOLD : 0);
        a.groups = kHeads; a.a_gs = uld * 2; a.w_gs = (long long)kHeadDim * uK * 2; a.c_gs = kHeadDim * 2; a.bias_gs = kHeadDim;
        if (u_split) a.a_k_dup = E;
        if (absorb_raw) {
            a.acc_init = (const float*)(tri ? pw + P.d_cc_kv : pw + P.d_in_kv) + E; a.acc_init_gs = kHeadDim;
            a.colsum = (const float*)(pw + P.c_in_kv) + E; a.colsum_gs = kHeadDim;
            a.stats_in = mr_u; a.stats_in_gs = (long long)rows_q * 2;
        }
        TP_TRY(launch(TP_F16, TP_F16, a, stream));
    } else
    TP_TRY(region_attention_launch(ws + W.q, kv_slab, kv_slab + kvE * 2, ws + W.o, B, g, s, stream, attn_mask, mask_mode,
                                   region_major ? 1 : 0));
    TP_TRY(mark());
    // 8. out_proj — folded into mlp[0] at pack time where fold_out_proj() says so (TP_TUNE_FOLD_OUT_PROJ)
    const bool fold = plan.fold;
    if (!fold) {
        GemmArgs a = plain_gemm(ws + W.o, E, pw + P.w_out, slab(W.a1), E, rows_q, E, E, (const float*)(pw + P.b_out), 0);
        TP_TRY(launch(TP_F16, TP_F16, a, stream));
    }
    TP_TRY(mark());
    // 9. mlp[0] + GELU   (on O with W_om = Wm0·Wout when folded)
    {
        GemmArgs a = fold ? plain_gemm(ws + W.o, E, pw + P.w_om, ws + W.a2, D, rows_q, D, E, (const float*)(pw + P.b_om), TP_LINEAR_GELU)
                          : plain_gemm(slab(W.a1), E, pw + P.w_m0, ws + W.a2, D, rows_q, D, E, (const float*)(pw + P.b_m0), TP_LINEAR_GELU);
        if (train) { a.flags |= TP_LINEAR_SAVE_PRE; a.C2 = ws + W.z2; }
        TP_TRY(launch(TP_F16, TP_F16, a, stream));
    }
    TP_TRY(mark());
    // 10. mlp[2] -> out
    {
        GemmArgs a = plain_gemm(ws + W.a2, D, pw + P.w_m2, out, D, rows_q, D, D, (const float*)(pw + P.b_m2),
                                0);
        TP_TRY(launch_maybe_splitk(TP_F16, desc->out_dtype, a, stream));
    }
    TP_TRY(mark());
    return TP_OK;
}
}  // namespace tp

extern "C" {

int tp_forward(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
               const int64_t xm_strides[3], const void* packed_weights, void* out, void* workspace,
               size_t workspace_bytes, void* stream) {
    return forward_impl(desc, x, x_strides, x_multi, xm_strides, packed_weights, out, workspace, workspace_bytes,
                        stream, nullptr, false);
}

int tp_forward_masked(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
                      const int64_t xm_strides[3], const void* packed_weights, void* out, void* workspace,
                      size_t workspace_bytes, const float* attn_mask, int mask_mode, void* stream) {
    if ((mask_mode != 0 && !attn_mask) || mask_mode < 0 || mask_mode > 2) { set_error("tp_forward_masked: bad mask / mask_mode"); return TP_ERR_INVALID_ARG; }
    return forward_impl(desc, x, x_strides, x_multi, xm_strides, packed_weights, out, workspace, workspace_bytes,
                        stream, nullptr, false, nullptr, mask_mode ? attn_mask : nullptr, mask_mode);
}

int tp_forward_parts(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* const xm_parts[4],
                     const int64_t part_strides[3], const void* packed_weights, void* out, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!xm_parts) { set_error("tp_forward_parts: xm_parts is NULL"); return TP_ERR_INVALID_ARG; }
    return forward_impl(desc, x, x_strides, nullptr, part_strides, packed_weights, out, workspace, workspace_bytes,
                        stream, nullptr, false, xm_parts);
}

int tp_forward_staged(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
                      const int64_t xm_strides[3], const void* packed_weights, void* out, void* workspace,
                      size_t workspace_bytes, void* stream, void* const* stage_events, int n_events) {
    if (!stage_events || n_events != TP_NUM_STAGES + 1) {
        set_error("tp_forward_staged: need %d events, got %d", TP_NUM_STAGES + 1, n_events);
        return TP_ERR_INVALID_ARG;
    }
    for (int i = 0; i < n_events; ++i)
        if (!stage_events[i]) { set_error("tp_forward_staged: event %d is NULL", i); return TP_ERR_INVALID_ARG; }
    return forward_impl(desc, x, x_strides, x_multi, xm_strides, packed_weights, out, workspace, workspace_bytes,
                        stream, stage_events, false);
}

int tp_release_stream(void* stream) { return release_stream_state((hipStream_t)stream); }

int tp_pack_forget(const void* packed) {
    if (!packed) { set_error("tp_pack_forget: NULL"); return TP_ERR_INVALID_ARG; }
    pack_registry_forget(packed);
    return TP_OK;
}

long long tp_debug_counter(int which) {
    if (which == TP_COUNTER_SIDE_STREAMS) return side_cache_size();
    if (which == TP_COUNTER_PAIR_LAUNCHES) return gemm_pair_launch_count();
    set_error("tp_debug_counter: unknown counter %d", which);
    return -1;
}

}  // extern "C"
