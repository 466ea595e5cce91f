// Internal declarations shared by the HIP translation units of libtokenpacker_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/tokenpacker.h"

namespace tp {

// ---- fixed geometry of the path (reference builder.py:40-49, 61, 67) --------------------------
constexpr int kEmbed = 1024;     // embed_dim == kv_dim == CLIP width
constexpr int kMulti = 4096;     // 4 CLIP layers concatenated
constexpr int kHeads = 8;
constexpr int kHeadDim = 128;

using bf16_t = __bf16;
using f16_t = _Float16;
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using f16x4 = _Float16 __attribute__((ext_vector_type(4)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x8_t = float __attribute__((ext_vector_type(8)));

template <typename T> struct Vec;
template <> struct Vec<bf16_t> { using x8 = bf16x8; using x4 = bf16x4; };
template <> struct Vec<f16_t> { using x8 = f16x8; using x4 = f16x4; };

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);      // hipGetLastError -> TP_ERR_LAUNCH

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives several GPUs must set it on
// each of them, and a transient failure must not be remembered (a function-local `static` / std::call_once set it on the
// first device only and cached a first failure for the life of the process).  One of these per kernel instantiation, as a function-local
// static: a bit per device ordinal (0 .. 127; beyond that the attribute is simply set before every launch — the call is cheap), set only
// after the runtime accepted the attribute there.  Two threads racing on a fresh device both set it — harmless.
struct DynLdsAttr {
    unsigned long long done[2] = {0ull, 0ull};
    hipError_t ensure(const void* kern, int lds_bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        const bool tracked = dev >= 0 && dev < 128;
        const unsigned long long bit = tracked ? 1ull << (dev & 63) : 0ull;
        if (tracked && (__atomic_load_n(&done[dev >> 6], __ATOMIC_ACQUIRE) & bit)) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e == hipSuccess && tracked) __atomic_fetch_or(&done[dev >> 6], bit, __ATOMIC_RELEASE);
        return e;
    }
};

// ---- tuning: the context of the call in progress on this thread, else the process-wide table (tp_api.hip) ----------------
int tuning(int key);
// Opened by every entry point that takes a tp_desc: tuning() reads desc->tuning (NULL: the process-wide table) until it closes.
struct TuningScope {
    explicit TuningScope(const tp_desc* d);
    ~TuningScope();
    TuningScope(const TuningScope&) = delete;
    TuningScope& operator=(const TuningScope&) = delete;
private:
    const void* prev_;
};

// ---- GEMM (tp_gemm.hip) -----------------------------------------------------------------------
// C[g][M,N] = epilogue(A[g][M,K] * W[g][N,K]^T) for g in [0, groups)
struct GemmArgs {
    const char* A; const char* W; char* C;
    const float* bias;
    const float* stats_in;            // LN_FOLD: per-row (mean, rstd) [M][2]
    // LN_FOLD on the 128-tile kernel (small M: every launch on the critical path counts): instead of (mean, rstd) from a
    // separate ln_finalize launch, the producing GEMM's (mean, M2) slabs [8][M][2] — merged per row by the kernel itself
    // with ln_merge_slabs(), the very code ln_finalize runs (same bits).  gemm_uses_small_kernel() tells the caller
    // whether a launch takes that route.
    const float* stats_parts; long long stats_parts_gs; float ln_inv_dim, ln_eps;
    const float* colsum; float* stats_out;
    const float* acc_init;            // optional fp32 [N]: initial value of the accumulators per output column (a constant row
                                      // vector added BEFORE the LayerNorm fold: rstd·(A·W^T + acc_init − mu·colsum) + bias)
    char* C2;                         // SAVE_PRE: pre-activation copy of C (same dtype / ldc), NULL otherwise
    const char* Z;                    // GELU_BWD: fp16 pre-activations [M, ldz] the result is multiplied by gelu'(.)
    long long ldz;                    // elements
    long long c2_gs, z_gs;            // per-group strides in bytes
    long long a_batch_stride_bytes;   // between batches of rows_per_batch rows
    long long lda_bytes;              // between rows inside a batch
    long long ldc;                    // elements
    long long ldw_bytes;              // between rows of W (0: K * 2 — a contiguous [N, K] weight)
    const char* A_parts[4];           // K split over 4 source tensors of k_part columns each (NULL: A alone)
    int k_part;
    int a_k_dup;                      // pair kernel and 128-tile kernel, contiguous A: the first a_k_dup K-elements of A (a multiple of 64) are
                                      // each used for TWO consecutive K-tiles of W, the rest once: K-tile j of the contraction reads A's K-tile
                                      // j / 2 while j < 2 w (w = a_k_dup / 64) and j - w after that, so K = a_k_dup + (A's width).  The absorbed
                                      // schedule's per-head V GEMM contracts u (E wide, a_k_dup = E) against W rows laid out as K-tile pairs
                                      // [hi_0 lo_0 hi_1 lo_1 ..] (K = 2 E): u·W_hi + u·W_lo with every byte of u fetched from HBM ONCE — the second
                                      // use of a K-tile follows the first by one K-tile and hits the L2.  0: off
    int parts_k_groups;               // A_parts + groups over K (128-tile kernel): group g covers K-tiles g*K/64 .. of the sources
    int tri;                          // statistics-only launches (NO_STORE): W is UPPER TRIANGULAR (W[n][k] = 0 for k < n): the output
                                      // tile at column n0 starts its K loop at K-tile n0 / 64 (tp_pack_qr.hip)
    int ln_second_moment;             // stats_parts consumers: the slabs are those of a triangular statistics GEMM — rstd from the
                                      // second moment (M2 + n mean^2), mean := 0
    int* tile_counters;               // persistent kernel: [groups][8] zeroed ints -> dynamic per-XCD tile queue (NULL: static)
    // "TT" mode (tt_rows > 0; tp_gemm8.hip AMODE 3): C[g][M,N] = sum over rows r of group g's range of A[r][m] W[r][n] —
    // both operands K-major, row strides lda_bytes / ldw_bytes; group g covers contraction rows g*K .. g*K + K - 1 of
    // tt_rows valid ones (rows beyond read as zero).  W rows may live in batches of tt_tpb K-tiles (64 rows each) with
    // w_batch_stride_bytes between batches (tt_bmagic = ceil(2^32 / tt_tpb); 0 = one batch), and its N columns may be
    // split over W_parts[4] tensors of n_part columns each.
    long long tt_rows;
    int tt_w_kcontig;                 // 1: only A is K-major; W is an ordinary K-contiguous [N, K] operand (ldw_bytes, w_gs)
    long long w_batch_stride_bytes;
    unsigned tt_bmagic; int tt_tpb;
    const char* W_parts[4];
    int n_part;
    // per-group strides (bytes for A/W/C, floats for the fp32 side arrays)
    long long a_gs, w_gs, c_gs, bias_gs, stats_in_gs, colsum_gs, stats_out_gs, acc_init_gs;
    int M, N, K;
    int m_begin, m_end;               // ping-pong kernel: tiles cover rows [m_begin, m_end) (m_end 0: M); row indices, the
                                      // bound M and every per-row array stay those of the whole problem
    int half_tiles;                   // ping-pong kernel: 128 x 256 tiles (tp_gemm8.hip HALF) instead of 256 x 256
    int tile192;                      // ping-pong kernel: 192 x 256 tiles (tp_gemm8.hip T192; plain launches over all rows)
    int rows_per_batch;
    int a_region_g, a_region_s;       // a_region_s > 0: the M rows are the fine tokens in REGION-MAJOR order (image, region in
                                      // raster order of the (g/s)^2 regions, key a*s + c inside the region): row r of the result
                                      // is computed from the raster token (qi*s + a)*g + qj*s + c of that image — the first K/V
                                      // layer reads the tower's rows in this order so that every later per-row tensor of the
                                      // K/V side holds a region's s*s tokens in consecutive rows (strided-A kernels only)
    // Region attention fused into the epilogues of the K and V in-projections (scale_factor 2: a region = 4 consecutive
    // rows of a region-major operand = one lane quad of the accumulator layout; tp_gemm_common.h attn_*_epilogue):
    //   attn_mode 1 (the K launch): nothing of K is stored; logit[h][r] = attn_scale * K[r, head h] . Q[r / 4, head h]
    //   attn_mode 2 (the V launch): O[r / 4, :] = sum over the region's 4 rows of softmax(logit)[r] * V[r, :] — C is O,
    //               fp16 [M / 4, ldc]
    // Output columns >= c_split_cols (a multiple of the tile width) go to a second slab: C + c_split_stride_bytes, column
    // index minus c_split_cols, same ldc — the first K/V layer leaves Hkv as a K slab and a V slab [rows, 1024] each instead
    // of interleaved [rows, 2048] halves (inference; its consumers then walk contiguous rows).  0: one slab.
    int c_split_cols; long long c_split_stride_bytes;
    int attn_mode;
    const char* attn_q;               // attn_mode 1: fp16 queries [M / 4, attn_ldq_bytes]
    long long attn_ldq_bytes;
    float* attn_logits;               // fp32 [N / 128][M]  (head-major)
    float attn_scale;
    // attn_decoupled (round 6; the centred chain only — every LayerNorm mean is 0 there): the K launch does not wait for the K/V
    // row statistics.  With mu = 0,  Q_h·K_r,h = rstd_r (Q_h·(Hkv_r Wcc^T + dcc)_h) + Q_h·b'_h  and the last term is the same for
    // the 4 keys of a region, i.e. softmax-invariant: the K launch writes the RAW dot product (attn_scale applied, nothing else)
    // and needs only Hkv and Q; the V launch multiplies the quad's logits by the K rows' rstd, which it finds in the MEAN slot
    // of its own (mean, rstd) pairs (ln_finalize_launch(pair_rstd) / attn_kstats_parts put it there — the slot is otherwise 0).
    // The statistics launch and the K launch are then independent and run side by side on two streams (tp_api.hip).
    int attn_decoupled;
    const float* attn_kstats_parts;   // V launch on the 128-tile kernel merging its producer's slabs (stats_parts): the K group's slabs
    int flags;                        // TP_LINEAR_*
    int groups;
    int tile;                         // 0 auto, 128, 256
    // Sticky fp16-saturation report: every fp16 epilogue clamps to +-65504 instead of producing inf; when sat_flag is given
    // (a device int32 the caller zeroed at some point), a wave that clamped anything ORs sat_bit into it.  One v_max3 per
    // two output elements and, for a tile that did saturate, one atomic — nothing otherwise.
    int* sat_flag; int sat_bit;
};
int gemm_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream);
int gemm_route_of(int in_dtype, int out_dtype, const GemmArgs& a);   // test hook: 0 small | 1 full | 2 half | 3 split | 4 192-row | 5 pair
bool gemm_uses_small_kernel(const GemmArgs& a);        // whether gemm_launch would run `a` on the 128-tile kernel (tp_gemm.hip)

// LayerNorm statistics of one row from the producing GEMM's NPARTS (mean, M2) slabs of 128 columns each ([NPARTS][M][2];
// M2 = sum of squared deviations from the slab's own mean): Chan's merge in slab order -> (mean, rstd) of nn.LayerNorm
// (biased variance).  All slabs are fetched before any is used (independent loads in flight).  Used by ln_finalize_kernel
// and by the 128-tile GEMM's LN-fold prologue: one arithmetic, identical bits.
// `second_moment`: the slabs describe y = R h + c~ of a triangular statistics GEMM (tp_pack_qr.hip), whose SECOND MOMENT is the
// variance wanted: M2 + n mean^2 (Chan's merge is exact algebra; every term is a sum of squares) -> (0, rstd).
template <int NPARTS>
__device__ __forceinline__ float2 ln_merge_values(const float2 (&st)[NPARTS], float inv_dim, float eps, bool second_moment = false) {
    float s1 = 0.f, q = 0.f, between = 0.f;
#pragma unroll
    for (int pp = 0; pp < NPARTS; ++pp) { s1 += st[pp].x; q += st[pp].y; }
    const float mu = s1 / (float)NPARTS;                        // slabs are equally sized (128 columns each)
#pragma unroll
    for (int pp = 0; pp < NPARTS; ++pp) { const float d = st[pp].x - mu; between = fmaf(d, d, between); }
    const float var = (q + 128.0f * between) * inv_dim;         // biased variance (nn.LayerNorm); >= 0 by construction
    if (second_moment) return make_float2(0.f, 1.0f / sqrtf(fmaf(mu, mu, var) + eps));
    return make_float2(mu, 1.0f / sqrtf(var + eps));
}
template <int NPARTS>
__device__ __forceinline__ float2 ln_merge_slabs(const float* __restrict__ pg, long long M, long long m, float inv_dim, float eps,
                                                 bool second_moment = false) {
    float2 st[NPARTS];
#pragma unroll
    for (int pp = 0; pp < NPARTS; ++pp) st[pp] = *(const float2*)(pg + ((long long)pp * M + m) * 2);
    return ln_merge_values<NPARTS>(st, inv_dim, eps, second_moment);
}

int gemm_pick_tile(int M, int N, int forced, int groups = 1);     // -> 128 or 256
// 256x256x64 ping-pong kernel (tp_gemm8.hip); gemm_launch routes tile-256 problems to it
int gemm8_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream);
// 256x128x64 pair kernel (tp_gemm_pair.hip): two co-resident 4-wave workgroups per CU, epilogues under the other's MFMAs
bool gemm_pair_supports(int in_dtype, int out_dtype, const GemmArgs& a);
int gemm_pair_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream);
long long gemm_pair_launch_count();
bool gemm_probes_built();           // -DTP_BUILD_PROBES (libtokenpacker_exp.so): the timing-probe instantiations TP_TUNE_PAIR_DEBUG selects exist
int gemm_pair_occupancy();
int gemm_pair_workgroups();                            // workgroups of a pair launch (two per CU)
int gemm8_persistent_cus();                            // workgroups of a persistent launch (CUs rounded down to 8)
constexpr int kMaxLaunches = 16;                       // GEMM launches of one forward that get tile-queue heads
constexpr size_t kCounterBytes = (size_t)kMaxLaunches * 64 * 4;
inline int gemm_stats_parts(int N) { return N / 128; }   // one (sum, sumsq) slab per 128 output columns

// ---- small kernels (tp_kernels.hip) -----------------------------------------------------------
// Activations between kernels are ALWAYS fp16 (see DESIGN.md "numerics"): `dtype` below is the
// element type of the caller's tensors (x, weights), outputs are fp16.
int point_queries_launch(int dtype, const void* x, const int64_t st[3], void* q0_f16, int B, int grid,
                         int s, hipStream_t stream);
// mask (optional, fp32): additive attn_mask — mask_mode 1: [s*s], 2: [(M B) 8, s*s] (batch index = region * B + image)
// region_major: K / V rows are in region-major order (GemmArgs::a_region_s) instead of the tower's raster order
int region_attention_launch(const void* q, const void* k, const void* v, void* o, int B,
                            int grid, int s, hipStream_t stream, const float* mask = nullptr, int mask_mode = 0,
                            int region_major = 0);                                                                      // fp16 in / fp16 out
// K/V in-projections absorbed into the query side (tp_kernels.hip): qt [B*M, 8, 1024], H2 k / v [B*N, 1024] fp16 with
// their per-row (mean, rstd) -> u [B*M, 8, 1024] fp16
int region_attention_absorbed_launch(const void* qt, const void* h2k, const void* h2v, const float* mr_k, const float* mr_v,
                                     void* u, int B, int grid, int s, hipStream_t stream, const float* mask = nullptr, int mask_mode = 0,
                                     // RAW form (q != NULL): h2k / h2v are the K / V halves of Hkv (row stride ld elements), qt was built
                                     // from Wc_k; q [B*M, 1024] fp16, d_k / c_k [1024] fp32 -> u normalised + mr_u [8][B*M][2]
                                     int ld = kEmbed, const void* q = nullptr, const float* d_k = nullptr, const float* c_k = nullptr,
                                     // u_split: qt arrives in fp32 (the round-4 / 5 form of the s >= 3 schedule); u [B*M, 8, E] fp16 either way
                                     float* mr_u = nullptr, bool u_split = false);
int pack_head_transpose_launch(const void* w_f16, void* dst_f16, hipStream_t stream);    // [8*128, 1024] -> [8][1024][128]
// tp_pack_qr.hip: W2 (fp16 [E,E]) and b2 (fp32 [E] or NULL) centred into matrix m of `scratch` (fp64), wbar [E+1] = the column means
// and mean(b2); Householder QR of the nmat matrices; R (fp16, upper triangular) and c~ (fp32) out; the centred chain weight
size_t pack_qr_scratch_bytes(int nmat);
int pack_qr_center_launch(const void* w2_f16, const float* b2, void* scratch, int m, float* wbar, hipStream_t stream);
int pack_qr_factor_launch(void* scratch, int nmat, hipStream_t stream);
int pack_qr_extract_launch(const void* scratch, int m, void* r_f16, float* ctil, hipStream_t stream, int* sat);
int pack_center_product_launch(const float* P, const float* c, const float* wbar, void* out_f16, const float* d, float* d_out,
                               hipStream_t stream, int* sat, const float* P2 = nullptr, void* out3_f16 = nullptr);   // out3: rows [hi_0 lo_0 .. hi_15 lo_15]
// lo = fp16(W' − fp16(W')), c_exact = rowsum(W'), d_exact = W'·v (v may be NULL), W' = w·diag(gamma) exact in fp32 (tp_kernels.hip)
int pack_ln_fold_residual_launch(int dtype, const void* w, const void* gamma, const float* v, void* lo_f16, float* c_exact,
                                 float* d_exact, int n_out, int n_in, hipStream_t stream);
bool absorb_kv(const tp_desc* desc, bool train);          // whether tp_forward runs the absorbed schedule for desc
bool fold_out_proj(const tp_desc* desc, bool train);      // whether out_proj is folded into mlp[0] for desc
int hd_slice_launch(const float* img, int H, int W, int h_block, int w_block, int h_res, int w_res, int hg, int wg,
                    float* crops, int block, hipStream_t stream);
int occupy_cus_launch(int blocks, int usec, int* sink, hipStream_t stream);
int hd_assemble_launch(const tp_hd_image* plan_host, int n_images, const void* tokens, const int32_t* crop_map, const void* sep,
                       const void* ret, void* out, int M, int D, hipStream_t stream);
int ln_finalize_launch(const float* parts, float* mean_rstd, long long M, int nparts, int groups, int ln_dim,
                       float eps, hipStream_t stream, bool second_moment = false, bool pair_rstd = false);
int pack_cast_f32_launch(int dtype, const void* src, float* dst, int n, hipStream_t stream);
// a batch of casts / copies of `dtype` sources in one launch (tp_kernels.hip: pack_batch_kernel)
enum { BATCH_OP_TO_F32 = 0, BATCH_OP_TO_F16 = 1, BATCH_OP_COPY16 = 2 };
struct BatchOp { const void* src; void* dst; long long n; int kind; int pad_; };
constexpr int kBatchOps = 24;
struct BatchOps {
    BatchOp op[kBatchOps]; int count;
    // (false: the table is full — the caller flushes with pack_batch_launch and starts over)
    bool add(int kind, const void* src, void* dst, long long n) {
        if (count >= kBatchOps) return false;
        op[count++] = BatchOp{src, dst, n, kind, 0};
        return true;
    }
};
int pack_batch_launch(int dtype, const BatchOps& ops, hipStream_t stream, int* sat);
// `sat` (optional): device int incremented once per element that did not fit fp16 and was clamped to +-65504
int pack_cast_f16_launch(int dtype, const void* src, void* dst_f16, long long n, hipStream_t stream, int* sat = nullptr);
int pack_transpose_f16_launch(const void* src_f16, void* dst_f16, int n, hipStream_t stream);          // [n,n]
int pack_round_f16_launch(const float* src, void* dst_f16, long long n, hipStream_t stream, int* sat = nullptr);   // saturating
int pack_bias_fold_launch(const void* w_f16, const float* v, const float* b, float* out, int n_out, int n_in,
                          hipStream_t stream);                                             // out = w·v + b  (b may be NULL)
int pack_ln_fold_launch(int dtype, const void* w, const void* bias, const void* gamma,
                        const void* beta, void* w_out, float* colsum, float* bias_out, int n_out,
                        int n_in, hipStream_t stream, int* sat = nullptr);
// debug: counts[i] += number of fp16 elements of buf[i] (n[i] of them) with |v| >= 65504 or NaN (a saturated epilogue)
int count_saturated_launch(const void* buf, long long n, int* count, hipStream_t stream);

// ---- packed-weight and workspace layouts (tp_api.hip) -----------------------------------------
struct PackedLayout {
    size_t w_kv0, b_kv0;          // [2048,4096] io dtype (consumed with the raw inputs), [2048] f32
    size_t w_kv2, b_kv2;          // [2][1024,1024] f16, [2][1024] f32
    size_t w_q1;                  // [1024,1024] f16
    size_t w_in_kv, c_in_kv, b_in_kv;   // LN-folded in-proj for k, v: [2][1024,1024] f16, [2][1024] f32 x2
    size_t w_in_q, c_in_q, b_in_q;      // LN-folded in-proj for q
    size_t w_c_kv, d_in_kv;       // fused LayerNorm chain: Wc = W'·W2 [2][1024,1024] f16, d = W'·b2 [2][1024] f32
    size_t w_c_q;                 // the same for the query side: W'q·Wq1 [1024,1024] f16 (q_proj_1 has no bias: d = 0)
    // triangular statistics (TP_TUNE_TRI_STATS, tp_pack_qr.hip): the chain with the mean folded away — Wc' = W'·W2c, d' = W'·b2c
    // (W2c / b2c: W2 / b2 centred over the output index), and the statistics GEMM's weight R (upper triangular, W2c = Q R) with
    // its constant c~ = Q^T b2c:  var = || R h + c~ ||^2 / E
    size_t w_cc_kv, d_cc_kv;      // [2][1024,1024] f16, [2][1024] f32
    size_t w_cc_q;                // [1024,1024] f16
    size_t w_qt_cc;               // per-head transposes of Wc'_k (absorbed schedule)
    size_t w_cc_v3;               // [E][2 E] f16: the rows of Wc'_v as K-tile pairs hi_t | lo_t (lo = what its fp16 rounding drops) — the absorbed
                                  // schedule's per-head V GEMM contracts the ONE fp16 u with it over K = 2 E (GemmArgs::a_k_dup: every K-tile of u serves its
                                  // hi_t | lo_t pair): u·W_hi + u·W_lo — the weight's rounding does not survive, u's does (round 5)
    size_t w_r_kv, c_r_kv;        // [2][1024,1024] f16 (zeros below the diagonal), [2][1024] f32
    size_t w_r_q;                 // [1024,1024] f16
    size_t wbar;                  // pack scratch: [3][1025] f32 column means of W2 (k, v, q) and the mean of b2 behind each
    size_t scratch_qr;            // pack scratch: three fp64 [1024][1025] matrices + Householder vectors
    size_t w_qt;                  // [8][1024][128] f16: per-head transposes of the LN-folded K in-proj (absorbed schedule)
    size_t w_qt_c;                // the same of Wc_k (absorbed schedule on the fused LayerNorm chain: the kernel walks Hkv)
    size_t w_out, b_out;          // [1024,1024] f16, [1024] f32
    size_t w_m0, b_m0;            // [D,1024] f16, [D] f32
    size_t w_m2, b_m2;            // [D,D] f16, [D] f32
    size_t w_om, b_om;            // out_proj folded into mlp[0]: (Wm0·Wout) [D,1024] f16, Wm0·bout + bm0 [D] f32
    size_t scratch_t, scratch_p;  // pack-time scratch: Wout^T [1024,1024] f16, the fp32 product [D,1024]
    size_t status;                // int32[64]: [0] = weight elements clamped to the fp16 range by tp_pack_weights,
                                  // [1] = 1 when w_om / b_om were built (TP_TUNE_FOLD_OUT_PROJ at pack time)
    size_t total;
};
PackedLayout packed_layout(int D);

// The schedule tp_forward runs for a descriptor under the tuning table OF THE MOMENT (tp_api.hip plan_schedule): one place
// decides; forward_impl, the workspace layout and the debug scan read it.  `masked`: the forward carries an attn_mask.
struct SchedulePlan {
    bool train;
    bool absorb, absorb_raw;       // K/V in-projections absorbed into the query side (scale_factor >= 3); on the fused LayerNorm chain
    bool u_split;                  // absorbed + centred chain: qt in fp32, the pre-multiplied V weight as hi + lo (round 4 also: u as hi | lo)
    bool fuse_ln;                  // plain schedule on the fused LayerNorm chain: H2 for its row statistics only
    bool fuse_q;                   // query side on the fused chain: Q1pre for its row statistics only
    bool region_major, fuse_attn;  // scale_factor 2: region-major K/V rows; attention inside the in-projections' epilogues
    bool fold;                     // out_proj folded into mlp[0]
    bool split_k;                  // TP_TUNE_SPLIT_K applies to this batch
    bool tri;                      // TP_TUNE_TRI_STATS: centred chain weights, triangular statistics GEMM, no mean anywhere
    bool decouple_k;               // fuse_attn on the centred chain: the K launch writes raw logits, independent of the row statistics (GemmArgs::attn_decoupled)
    bool need_h2, need_kv, need_q1pre, need_a1;   // workspace slabs the schedule writes
};
SchedulePlan plan_schedule(const tp_desc* desc, bool train, bool masked);

constexpr size_t kNoSlab = ~(size_t)0;             // offset of a slab the schedule does not have (never dereferenced)
struct WorkspaceLayout {
    size_t status;                // 256 B at offset 0: int32[0] = sticky fp16-saturation bits (bit k: stage k - 1 of tp_forward_staged, bit 0: query side)
    size_t q0, hkv, h2, stats_kv, mr_kv, kv, q1pre, stats_q, mr_q, q, o, a1, a2;
    size_t attn_aux;              // logits [8][B*N] fp32 (attention in the in-projection epilogues) | (e/a, a) [8][B*M][2] (absorbed, RAW)
    size_t counters;              // zeroed once per forward: tile-queue heads of the persistent GEMM launches
    size_t splitk;                // small batches: fp32 partial results of a K-split GEMM (TP_TUNE_SPLIT_K), kSplitKBytes
    size_t z1, z2;                // training forward only: fp16 pre-GELU activations [B*N, 2048], [B*M, D]
    size_t total;
    int stats_parts_kv, stats_parts_q;
};
// Schedule-aware: a slab the plan does not write takes no space (kNoSlab).  At B = 256, scale_factor 2, D = 4096 the default
// inference schedule needs 1.24 GB (no H2, no K | V, no Q1pre); the training layout keeps every slab the backward reads.
WorkspaceLayout workspace_layout(int B, int grid, int s, int D, const SchedulePlan& plan);
WorkspaceLayout workspace_layout(int B, int grid, int s, int D, bool train = false);    // the plan of (desc-less) defaults: training = every slab
// K-split of a latency-bound GEMM (TP_TUNE_SPLIT_K): S * tiles <= 512 workgroups of 128 x 128 (two per CU) -> at most
// 512 * 128 * 128 fp32 partial values
constexpr size_t kSplitKBytes = (size_t)512 * 128 * 128 * 4;
// out[m, n] = epilogue(sum over the S partials [S][M][N] fp32, in split order): + bias, optional erf GELU, cast to out_dtype
int splitk_reduce_launch(const float* partials, int S, int M, int N, const float* bias, int gelu, void* out, long long ldc,
                         int out_dtype, hipStream_t stream, int split_cols = 0, long long split_stride_elems = 0,
                         int* sat_flag = nullptr, int sat_bit = 0);       // fp16 output: the sticky saturation report (GemmArgs::sat_flag)

// ---- backward helpers (tp_bwd.hip) -------------------------------------------------------------
int bw_transpose_launch(int src_dtype, int dst_dtype, const void* src, long long ld, int rows_per_batch,
                        long long batch_stride, int R, int C, void* dst, long long ldd, int Rpad, const float* mean_rstd,
                        const float* gamma, const float* beta, float* colsum_part, hipStream_t stream);
// `out_scale` (device, optional): the sum is multiplied by out_scale[0] before the cast — the 1 / S of a backward chain that carries
// its gradients scaled by a power of two S (bw_scale_from_partials_launch)
// a batch of plain [R, C] -> [C, Rpad] transposes (+ cast) in one launch (tp_bwd.hip: transpose_batch_kernel)
struct TransposeOp { const void* src; void* dst; long long ld, ldd; int R, C, Rpad, pad_; };
constexpr int kTransposeBatch = 8;
struct TransposeBatch {
    TransposeOp op[kTransposeBatch]; int count;
    bool add(const void* src, long long ld, int R, int C, void* dst, int Rpad) {
        if (count >= kTransposeBatch) return false;
        op[count++] = TransposeOp{src, dst, ld, (long long)Rpad, R, C, Rpad, 0};
        return true;
    }
};
int bw_transpose_batch_launch(int src_dtype, int dst_dtype, const TransposeBatch& tb, hipStream_t stream);
int bw_reduce_parts_launch(int dst_dtype, const float* part, long long part_stride, int nparts, long long n, void* out,
                           hipStream_t stream, const float* out_scale = nullptr);
// Dynamic power-of-two scale of a gradient tensor (the backward of a bf16 model runs its chain in fp16 — 11-bit mantissas, the
// saved fp16 activations read in place by the weight gradients — and fp16's range needs the incoming dy brought to a known
// magnitude): bw_scale_from_partials_launch turns the per-workgroup max |dy| a column-sum pass left behind into scale[0] = S = 2^k with
// amax * S in [16, 32) and scale[1] = 1 / S (S = 1 for an all-zero or non-finite tensor); bw_scale_cast_launch writes
// dst_f16 = saturate(src * S).  Nothing here synchronises: S lives on the device.
int bw_scale_cast_launch(int src_dtype, const void* src, long long n, const float* scale, void* dst_f16, hipStream_t stream,
                         int* sat_flag = nullptr);
// Weight gradient straight from the row-major activations (tp_gemm8.hip, K-major operands):
//   dW[Nout, Kin] = sum over the R token rows of dY[r, n] * X[r, k],   split over the rows, fp32 partials -> out_dtype.
// X may be batch-strided (rows_per_batch % 64 == 0) and its Kin columns split over four tensors of n_part columns.
struct WgradX {
    const void* x; long long ldx;                     // elements
    int rows_per_batch; long long batch_stride;       // 0 / 0: contiguous rows
    const void* const* parts; int n_part;             // NULL / 0: one tensor
};
bool wgrad_tt_supported(long long R, int Nout, int Kin, const WgradX& X, long long ldy);
// rows [0, split_row) of dW go to grad_out, rows [split_row, Nout) to grad_out_hi (split_row = 0: all to grad_out)
int wgrad_tt_launch(int dtype, const void* dY, long long ldy, const WgradX& X, long long R, int Nout, int Kin,
                    float* part, size_t part_bytes, int out_dtype, void* grad_out, int* counters, hipStream_t stream,
                    void* grad_out_hi = nullptr, int split_row = 0, const float* out_scale = nullptr);
size_t wgrad_tt_part_bytes(int Nout, int Kin);
int wgrad_tn_launch(int dtype, const void* dY, long long ldy, const void* XT, long long rpad, long long R, int Nout, int Kin,
                    float* part, size_t part_bytes, int out_dtype, void* grad_out, int* counters, hipStream_t stream,
                    const float* out_scale = nullptr);

constexpr int kColsumMaxSlices = 512;              // row slices of the column-sum kernel: scratch = slices * C floats
// amax_part / amax_count (optional): the pass also leaves max |v| per workgroup (*amax_count entries, <= kColsumAmaxParts) — the
// incoming dy is read ONCE for its bias gradient and for the dynamic scale of the fp16 gradient chain (bw_scale_from_partials_launch)
constexpr int kColsumAmaxParts = 8 * kColsumMaxSlices;
int bw_colsum_rows_launch(int dtype, const void* src, long long ld, long long R, int C, float* part, hipStream_t stream,
                          float* amax_part = nullptr, int* amax_count = nullptr);
int bw_scale_from_partials_launch(const float* amax_part, int nparts, float* scale, hipStream_t stream);
constexpr int kReduceSlices = 32;                  // stage-1 slices of the many-parts reduction: scratch = kReduceSlices * n floats
int bw_reduce_many_parts_launch(int dst_dtype, const float* part, long long part_stride, int nparts, int n, void* out,
                                float* scratch, hipStream_t stream, const float* out_scale = nullptr);
// part: [nblocks][3][E] — dgamma, dbeta, colsum(dx) partials.  beta / xn (optional): also write the LayerNorm's output in gdtype
int bw_ln_backward_launch(int gdtype, const void* dy, const void* x_f16, const float* mean_rstd, const float* gamma,
                          void* dx, float* part, int nblocks, long long rows, hipStream_t stream, const float* beta = nullptr,
                          void* xn = nullptr, int* sat_flag = nullptr);
int bw_region_attention_launch(int gdtype, const void* q, const void* k, const void* v, const void* dout, void* dq,
                               void* dk, void* dv, int B, int grid, int s, hipStream_t stream, int* sat_flag = nullptr);
int validate_desc(const tp_desc* d);
void pack_registry_put(const void* packed, const tp_desc* d, bool train_pack);      // tp_api.hip: what an image was packed for
int pack_registry_check(const void* packed, const tp_desc* d, bool train);
void pack_registry_forget(const void* packed);
GemmArgs plain_gemm(const void* A, long long lda_elems, const void* W, void* C, long long ldc, int M, int N, int K,
                    const float* bias, int flags);
long long max_images_per_launch(const tp_desc* desc);
int forward_impl(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
                 const int64_t xm_strides[3], const void* packed_weights, void* out, void* workspace,
                 size_t workspace_bytes, void* stream_, void* const* stage_events, bool train,
                 const void* const* xm_parts = nullptr, const float* attn_mask = nullptr, int mask_mode = 0);

}  // namespace tp
