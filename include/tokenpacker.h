/*
 * tokenpacker.h — C ABI of libtokenpacker_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path: the TokenPacker region-to-point visual projector of
 * CircleRadon/TokenPacker, reference `llava/model/multimodal_projector/builder.py:39-137`
 * (class TokenPacker; called from `llava/model/llava_arch.py:97`).  The reference has no FFI of
 * its own (it is 100 % eager PyTorch); the entry points below are what a binding for this path
 * has to offer, and each one names the reference code it replaces.  The Python side
 * (`tokenpacker_amd/_capi.py`) binds them with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - Plain C types only: device pointers, sizes, element strides, an opaque stream handle
 *    (`hipStream_t` passed as `void*`; NULL = the legacy default stream).
 *  - The caller owns every buffer.  The library allocates nothing on the device; its only global state is a
 *    thread-local error string, the tuning table of tp_set_tuning() and, per caller stream, one side stream +
 *    event pair that tp_forward forks from / joins back into `stream` for the query side of the path.
 *  - Every call only ENQUEUES work (on `stream` and that side stream); it never synchronises the device.
 *  - Return value: 0 on success, a negative tp_status otherwise; tp_last_error() describes the
 *    failure for the calling thread.  Nothing throws across the ABI.
 *  - All tensors are row-major.  `dtype` is the element type of activations AND weights.
 */
#ifndef TOKENPACKER_H_
#define TOKENPACKER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 5 (round 6): tp_linear_args.reserved1 became `ldw` and the packed-weight image's layout changed (w_cc_v3: interleaved
 * hi_t | lo_t K-tile pairs) — both already in round 5 under a stale 4 —, TP_TUNE_DECOUPLE_K and TP_TUNE_BWD_CHAIN (18 knobs), the probe instantiations
 * and test hooks left the product library (tokenpacker_test.h / libtokenpacker_exp.so), tp_debug_counter. */
#define TP_ABI_VERSION 5

typedef enum tp_status {
    TP_OK = 0,
    TP_ERR_INVALID_ARG = -1,   /* NULL pointer, bad shape, unsupported dtype ...                  */
    TP_ERR_BAD_SCALE = -2,     /* raw_grid % scale_factor != 0 (reference builder.py:51-52)       */
    TP_ERR_WORKSPACE = -3,     /* workspace / packed buffer too small                             */
    TP_ERR_LAUNCH = -4         /* HIP reported an error when enqueueing                           */
} tp_status;

typedef enum tp_dtype {
    TP_BF16 = 0,
    TP_F16 = 1,
    TP_F32 = 2                 /* only valid as tp_desc.out_dtype (fp32-output validation mode)   */
} tp_dtype;

/* Problem descriptor.  Mirrors the constructor arguments of the reference module
 * (builder.py:40-49: raw_grid=24, embed_dim=1024, num_heads=8, kv_dim=1024, hidden_size,
 * scale_factor) plus the batch.  embed_dim / kv_dim / the 4096-wide multi-level input are fixed
 * by the reference (builder.py:61,67 hard-code nn.Linear(4096,1024)). */
typedef struct tp_desc {
    int32_t batch;         /* B: images or HD crops                                               */
    int32_t raw_grid;      /* g: 24 for CLIP-L/14 @ 336 px; tokens per image N = g*g              */
    int32_t scale_factor;  /* s: must divide raw_grid; M = (g/s)^2 coarse queries                 */
    int32_t hidden_size;   /* D: LLM width (multiple of 128)                                      */
    int32_t dtype;         /* tp_dtype of x, x_multi and all weights: TP_BF16 or TP_F16           */
    int32_t out_dtype;     /* tp_dtype of `out`: == dtype, or TP_F32                              */
    float   ln_eps;        /* 1e-6 (builder.py:48)                                                */
    int32_t flags;         /* 0 | TP_DESC_TRAIN_PACK (tp_pack_weights) | TP_DESC_MASKED (sizes of a masked forward); other bits 0 */
    const struct tp_tuning* tuning;    /* ABI 4: the tuning context this call reads its knobs from; NULL: the process-wide table */
} tp_desc;
/* tp_pack_weights only: the image will serve tp_forward_train / tp_backward — the weights that only the inference
 * schedules read (Wc / d of the fused LayerNorm chain, the per-head transposes of the absorbed schedule, the out_proj fold)
 * are not built.  A training step re-packs every step (the parameters just changed) and must not pay for them. */
#define TP_DESC_TRAIN_PACK 1
/* tp_workspace_bytes / tp_forward_masked: the forward carries an attn_mask.  The mask-less scale_factor-2 schedule never
 * writes K | V (attention runs in the in-projections' epilogues) and its workspace has no room for them; a masked forward
 * runs the plain schedule, which does — size its workspace with this flag set (tp_forward_masked checks). */
#define TP_DESC_MASKED 2

/* The 23 tensors of the reference state dict, all of element type desc.dtype,
 * each contiguous, nn.Linear layout [out_features, in_features]. */
typedef struct tp_weights {
    const void* q_proj_1_weight;        /* [1024,1024]  builder.py:59 (no bias)                   */
    const void* k_proj_1_0_weight;      /* [1024,4096]  builder.py:61                             */
    const void* k_proj_1_0_bias;        /* [1024]                                                 */
    const void* k_proj_1_2_weight;      /* [1024,1024]  builder.py:64                             */
    const void* k_proj_1_2_bias;        /* [1024]                                                 */
    const void* v_proj_1_0_weight;      /* [1024,4096]  builder.py:67                             */
    const void* v_proj_1_0_bias;
    const void* v_proj_1_2_weight;      /* [1024,1024]  builder.py:70                             */
    const void* v_proj_1_2_bias;
    const void* ln_q_1_weight;          /* [1024]       builder.py:73                             */
    const void* ln_q_1_bias;
    const void* ln_k_1_weight;          /* [1024]       builder.py:74                             */
    const void* ln_k_1_bias;
    const void* ln_v_1_weight;          /* [1024]       builder.py:75                             */
    const void* ln_v_1_bias;
    const void* clip_attn_in_proj_weight;   /* [3072,1024]  builder.py:77 (q,k,v stacked)         */
    const void* clip_attn_in_proj_bias;     /* [3072]                                             */
    const void* clip_attn_out_proj_weight;  /* [1024,1024]                                        */
    const void* clip_attn_out_proj_bias;    /* [1024]                                             */
    const void* mlp_0_weight;           /* [D,1024]     builder.py:79                             */
    const void* mlp_0_bias;             /* [D]                                                    */
    const void* mlp_2_weight;           /* [D,D]        builder.py:82                             */
    const void* mlp_2_bias;             /* [D]                                                    */
} tp_weights;

/* ---- library info ---------------------------------------------------------------------------- */
int         tp_version(void);                 /* == TP_ABI_VERSION                                 */
const char* tp_last_error(void);              /* thread-local, never NULL                          */

/* ---- sizes ----------------------------------------------------------------------------------- */
/* Bytes of the packed-weight buffer for (hidden_size, dtype); 0 on invalid arguments. */
size_t tp_packed_weight_bytes(const tp_desc* desc);
/* Byte offset, inside the packed-weight buffer, of an int32[64] status block written by tp_pack_weights():
 * [0] = number of weight elements that did not fit the fp16 range the kernels keep every post-first-layer weight in
 *       and were clamped to +-65504 (a bf16 model can hold such values; results would silently differ from the
 *       reference, so the Python wrapper raises OverflowError when it is non-zero),
 * [1] = 1 when the out_proj∘mlp[0] fold was built (TP_TUNE_FOLD_OUT_PROJ set at pack time),
 * [2] = 1 when the fused LayerNorm chain's Wc = W'·W2 / d = W'·b2 were built (TP_TUNE_FUSE_KV_LN at pack time).
 * 0 on invalid args. */
size_t tp_packed_status_offset(const tp_desc* desc);
/* Bytes of scratch tp_forward() needs for this descriptor (depends on batch); 0 on invalid args. */
size_t tp_workspace_bytes(const tp_desc* desc);
/* The first TP_WORKSPACE_STATUS_BYTES of every workspace are a STATUS BLOCK the caller zeroes once (after allocating):
 *   int32[0]  sticky fp16-saturation bits.  Every activation between the kernels is fp16 and every epilogue CLAMPS to +-65504
 *             where the reference's fp16 / bf16 arithmetic would have produced inf or a larger finite value; an inference
 *             forward that clamps anything ORs bit k into this word (k = 1 + index of the stage in tp_forward_staged's list,
 *             bit 0 = the query side).  The library never clears it: read it back whenever convenient (a 4-byte copy),
 *             nonzero = some forward since the last clear differs from what the reference would have computed.
 *             Covered: every fp16 epilogue of the GEMM kernels and the K-split reduction of small batches (ABI 4).  NOT covered: a
 *             NaN inside a GEMM epilogue (the running max that tracks |v| is a v_max3, which drops NaN operands; the K-split
 *             reduction does report NaN), and training forwards (their epilogues have no register to spare) — scan those
 *             with tp_debug_count_saturated, which counts values at the bound AND NaNs. */
#define TP_WORKSPACE_STATUS_BYTES 256

/* ---- one-time weight preparation -------------------------------------------------------------
 * Replaces what `TokenPacker.__init__` + `load_state_dict` leave in the nn.Parameters
 * (builder.py:59-83, llava_arch.py:78-83): re-lays the 23 tensors out for the kernels — K/V first
 * layers concatenated into one [2048,4096] operand, the three LayerNorm affines folded into the
 * attention in-projection (W' = W·diag(gamma), c = rowsum(W'), b' = W·beta + b), out_proj folded into
 * mlp[0] (W = Wm0·Wout, b = Wm0·bout + bm0), biases widened to fp32.  Must be re-run whenever a parameter changes.  `packed` needs tp_packed_weight_bytes(). */
int tp_pack_weights(const tp_desc* desc, const tp_weights* raw, void* packed, size_t packed_bytes,
                    void* stream);
/* The library keeps a HOST-side note of what tp_pack_weights wrote at an address (hidden size, dtype, TP_DESC_TRAIN_PACK) and
 * refuses a forward the image cannot serve.  The note describes the address, not the bytes: before the memory of an image is
 * freed / reused for anything but a re-pack in place, drop it with tp_pack_forget (an unknown address is accepted unchecked). */
int tp_pack_forget(const void* packed);

/* ---- the hot path ------------------------------------------------------------------------------
 * Replaces `TokenPacker.forward((x, x_multi))` (builder.py:107-137).
 *   x        [B, g*g, 1024]  element strides x_strides[3]  (innermost stride must be 1)
 *   x_multi  [B, g*g, 4096]  element strides xm_strides[3] (innermost stride must be 1)
 *            — strides because the CLIP tower hands over `[:, 1:]` slices of CLS-prefixed
 *              buffers (clip_encoder.py:37-38,62); base pointers must be 16-byte aligned and row
 *              strides multiples of 8 elements.
 *   out      [B, M, D] contiguous, element type desc.out_dtype.
 * `attn_mask` of the reference signature is always None on the real path (llava_arch.py:97): tp_forward is the
 * mask-less call, tp_forward_masked below honours one.
 *
 * Schedules (same function, same parity gates; what differs is which intermediates exist):
 *   training (tp_forward_train)     : every layer as its own GEMM — the backward needs H2, K and V;
 *   inference, scale_factor 2       : fused LayerNorm chain (TP_TUNE_FUSE_KV_LN): the K/V second layer is computed for its
 *                                     LayerNorm statistics only, the in-projection reads Hkv through Wc = W'·W2, and region
 *                                     attention runs in the epilogues of those two GEMMs (TP_TUNE_FUSE_ATTN): K, V are not written;
 *   inference, scale_factor >= 3    : K/V in-projections absorbed into the query side (TP_TUNE_ABSORB_KV, see
 *                                     tp_region_attention_absorbed) — on the fused LayerNorm chain too: the second K/V layer is
 *                                     computed for its statistics only and the attention kernel walks Hkv. */
int tp_forward(const tp_desc* desc,
               const void* x, const int64_t x_strides[3],
               const void* x_multi, const int64_t xm_strides[3],
               const void* packed_weights,
               void* out,
               void* workspace, size_t workspace_bytes,
               void* stream);

/* tp_forward with the `attn_mask` argument of `TokenPacker.forward(x, attn_mask)` (builder.py:107,130), as
 * nn.MultiheadAttention applies it: an ADDITIVE fp32 mask on the scaled logits (a boolean mask is the caller's
 * 0 / -inf).  mask_mode 1: attn_mask [s*s] — the reference's 2-D (L=1, S=s*s) form, shared by every image, region
 * and head; mask_mode 2: attn_mask [(M*B)*8, s*s] — its 3-D (N*num_heads, L=1, S) form with the reference's batch
 * index N = region * B + image (divide_feature's order, builder.py:96-105); mask_mode 0: none.  Key index inside a
 * region: a*s + b for fine token (i*s + a, j*s + b).  Inference only. */
int tp_forward_masked(const tp_desc* desc,
                      const void* x, const int64_t x_strides[3],
                      const void* x_multi, const int64_t xm_strides[3],
                      const void* packed_weights, void* out, void* workspace, size_t workspace_bytes,
                      const float* attn_mask, int mask_mode, void* stream);

/* tp_forward with x_multi given as its FOUR [B, g*g, 1024] sources — the CLIP hidden states 12, 16, 22, 23 that
 * `CLIPVisionTower.feature_select` concatenates (clip_encoder.py:28-32) — each with the element strides
 * `part_strides` (e.g. the `[:, 1:]` slice of a [B, 577, 1024] hidden state).  The first GEMM walks the four
 * sources as K-ranges, so the tower's torch.cat (1.2 GB written and read again at B = 256) never happens.
 * Bit-identical to tp_forward on the concatenated tensor. */
int tp_forward_parts(const tp_desc* desc,
                     const void* x, const int64_t x_strides[3],
                     const void* const xm_parts[4], const int64_t part_strides[3],
                     const void* packed_weights,
                     void* out,
                     void* workspace, size_t workspace_bytes,
                     void* stream);

/* Same as tp_forward, additionally recording caller-owned HIP events (hipEvent_t, created with
 * timing enabled) on `stream` at the TP_NUM_STAGES+1 stage boundaries, so a benchmark can time each
 * kernel of the schedule inside the real forward (bench.py's `roofline` object).  Stages, in order:
 * point_queries, kv_layer0(+GELU), kv_layer2(+stats), kv_inproj(LN-fold), q_proj_1(+stats),
 * q_inproj(LN-fold), region_attention, out_proj, mlp0(+GELU), mlp2.
 * (The staged forward runs on ONE stream.  With attention in the in-projections' epilogues — TP_TUNE_FUSE_ATTN, the
 * scale_factor-2 default — the K launch needs Q: the whole query side then runs inside the first stage, the two q stages are
 * empty, `kv_inproj` is the K launch (logits) and `region_attention` the V launch (softmax-weighted sums -> O).) */
#define TP_NUM_STAGES 10
int tp_forward_staged(const tp_desc* desc,
                      const void* x, const int64_t x_strides[3],
                      const void* x_multi, const int64_t xm_strides[3],
                      const void* packed_weights,
                      void* out,
                      void* workspace, size_t workspace_bytes,
                      void* stream,
                      void* const* stage_events, int n_events);

/* ---- per-kernel entry points (used by the parity tests; same kernels tp_forward launches) ---- */

/* NUMERICS NOTE for the per-kernel entry points: activations BETWEEN kernels are always fp16,
 * whatever desc.dtype is (bf16 models: weights/inputs widen to fp16 exactly; keeping fp16
 * intermediates is what brings the bf16 path within 1e-3 of exact math, DESIGN.md "Numerics").
 *
 * Coarse point queries: fp32 bilinear (align_corners=False) g*g -> (g/s)^2, rounded once to
 * desc.dtype (the reference's cast, builder.py:117-118) and stored as fp16.
 * x has element type desc.dtype; q0 is fp16 [B, M, 1024] contiguous. */
int tp_point_queries(const tp_desc* desc, const void* x, const int64_t x_strides[3], void* q0,
                     void* stream);

/* Region-to-point attention core on projected tensors (post in-proj, pre out-proj), all fp16:
 *   q [B, M, 1024], k and v [B, g*g, 1024] contiguous; o [B, M, 1024].
 * 8 heads x d=128, logits scaled by 1/sqrt(128), softmax over the s*s tokens of the query's own
 * region.  Replaces divide_feature (builder.py:96-105, 122-124) + the bmm/softmax/bmm of
 * nn.MultiheadAttention (builder.py:126-130; torch/nn/functional.py:6576-6594). */
int tp_region_attention(const tp_desc* desc, const void* q, const void* k, const void* v, void* o,
                        void* stream);

/* The same attention with the K/V in-projections ABSORBED into the query side — what tp_forward runs for
 * scale_factor >= 3 (TP_TUNE_ABSORB_KV): with one query per region, Q_h·(W'k n_t + b'k)_h = (W'k_h^T Q_h)·n_t + const and
 * sum_t p_t (W'v n_t + b'v)_h = W'v_h (sum_t p_t n_t) + b'v_h, so the two [B·576,1024]x[1024,1024] in-projection GEMMs
 * shrink to query-sized work (1/s^2).  All fp16:
 *   qt [B, M, 8, 1024] = per-head Q_h · W'k_h (W' = LayerNorm-folded in_proj rows of k),
 *   h2k / h2v [B, g*g, 1024] = the PRE-LayerNorm second-layer outputs, normalised on load with
 *   mr_k / mr_v [B*g*g][2] = their per-row (mean, rstd) from tp_ln_finalize;
 *   u [B, M, 8, 1024] = per head sum_t softmax_t(qt_h·n^k_t / sqrt(128)) n^v_t   (then O_h = u_h·W'v_h^T + b'v_h).
 * s*s <= 64. */
int tp_region_attention_absorbed(const tp_desc* desc, const void* qt, const void* h2k, const void* h2v,
                                 const float* mr_k, const float* mr_v, void* u, void* stream);

/* Generic fused linear used for every dense contraction of the path (11 nn.Linear calls of
 * builder.py:112,113,120,126-130,136):  C[M,N] = epilogue(A[M,K] · W[N,K]^T).
 * flags: TP_LINEAR_* below.  `bias` fp32 [N] or NULL.  With TP_LINEAR_LN_FOLD the epilogue applies
 * a LayerNorm that precedes the linear: C = rstd_m·(acc − mu_m·colsum_n) + bias_n, with
 * `row_mean_rstd` = fp32 [M][2] (mean, rstd) per row of A, produced by tp_ln_finalize().
 * With TP_LINEAR_ROW_STATS the kernel writes, per 128 output columns, the (mean, M2) of ITS rounded output — M2 = sum of
 * squared deviations from that slab's own mean; merged by Chan's formula, accurate even when |mean| >> std — to
 * `row_stats_out` ([N/128][M][2]: one slab per 128 output columns; tp_linear_stats_parts() returns N/128).
 * One call addresses at most 4 GiB of output, (M + 256) * ldc * sizeof(element) < 2^32 (the stores go through a
 * range-checked 32-bit buffer descriptor); larger problems are rejected with TP_ERR_INVALID_ARG.  tp_forward /
 * tp_forward_parts split a batch beyond that bound (about 1800 images at hidden_size 4096) into consecutive chunks
 * themselves — no operation mixes images, the result is bit-identical; the training entry points reject it. */
enum {
    TP_LINEAR_GELU = 1,        /* exact erf GELU after bias (nn.GELU(), builder.py:63,69,81)      */
    TP_LINEAR_LN_FOLD = 2,
    TP_LINEAR_ROW_STATS = 4,
    TP_LINEAR_OUT_F32 = 8,     /* retired: use tp_linear_args.out_dtype = TP_F32                    */
    TP_LINEAR_SAVE_PRE = 16,   /* with GELU: also store the pre-activation (training forward)       */
    TP_LINEAR_GELU_BWD = 32,   /* multiply the result by gelu'(z), z = fp16 pre-activations (backward) */
    TP_LINEAR_NO_STORE = 64    /* with ROW_STATS: compute the row statistics of the (unrounded) result and store NOTHING else
                                  — C may be NULL (the LayerNorm statistics of a tensor nobody needs to materialise)     */
};
typedef struct tp_linear_args {
    int32_t M, N, K;           /* N % 128 == 0, K % 64 == 0                                        */
    int32_t dtype;             /* element type of A and W: TP_BF16 / TP_F16                        */
    int32_t out_dtype;         /* element type of C: TP_BF16 / TP_F16 / TP_F32                     */
    int32_t flags;
    int32_t rows_per_batch;    /* A row r lives at A + (r / rpb)*a_batch_stride + (r % rpb)*lda   */
    int32_t a_k_dup;           /* (ABI 5; was reserved0) 0, or a multiple of 64 <= K / 2: the first a_k_dup K-elements of A are each used for
                                * TWO consecutive 64-wide K-tiles of W, the rest once (A is K - a_k_dup wide) — the per-head V GEMM of the absorbed
                                * schedule (u against weight rows laid out as K-tile pairs hi_t | lo_t).  Contiguous A, N % 128 == 0; served by the
                                * pair kernel or the 128-tile kernel by launch size (TP_TUNE_PAIR_GEMM), bit-identical either way */
    int64_t a_batch_stride;    /* elements; ignored when rows_per_batch >= M                       */
    int64_t lda;               /* elements between consecutive rows of A                           */
    int64_t ldc;               /* elements between consecutive rows of C                           */
    const void* A;
    const void* W;             /* [N,K], rows ldw elements apart (ldw = 0: contiguous)             */
    const float* bias;
    void* C;
    const float* row_mean_rstd;/* LN_FOLD: fp32 [M][2]                                             */
    const float* colsum;       /* LN_FOLD: fp32 [N]                                                */
    int32_t tile;              /* 0 = auto, 128 or 256: force the block tile                       */
    int32_t ldw;               /* elements between consecutive rows of W; 0 = K (contiguous).  (ABI 5; round 5: the field was
                                * reserved1.  It exists for tools/stride_ab.py: an isolated LDS-DMA stream over rows a power of two
                                * apart runs at 28 GB/s per CU against 123 with +64 elements of padding — tools/probes/
                                * operand_fetch_probe.hip — but INSIDE the GEMMs, where an XCD's 32 tiles share their operands,
                                * padded A / W / C strides measure +-0.5 %, profiles/r05f_stride_ab.json; nothing is padded) */
    float*  row_stats_out;     /* ROW_STATS                                                        */
} tp_linear_args;
int tp_linear(const tp_linear_args* args, void* stream);
/* (mean, M2) slabs [parts][M][2] written by a TP_LINEAR_ROW_STATS call -> per-row (mean, rstd)
 * [M][2] of nn.LayerNorm(ln_dim, eps) (biased variance), for a TP_LINEAR_LN_FOLD call.  ln_dim == parts * 128. */
int tp_ln_finalize(const float* row_stats, int parts, int64_t M, int ln_dim, float eps,
                   float* row_mean_rstd, void* stream);
/* Number of row-stat slabs a TP_LINEAR_ROW_STATS call with these M,N (and args->tile) writes. */
int tp_linear_stats_parts(const tp_linear_args* args);

/* ---- training: forward that keeps the backward's operands, and the backward pass ---------------------------
 * Stage-1 training of the reference updates ONLY the projector (llava/train/train.py:950-953); the CLIP features
 * come from a frozen, no_grad tower (clip_encoder.py:46).  tp_backward therefore produces the gradients of the
 * 23 parameters (tp_grads mirrors tp_weights, each tensor contiguous, element type desc.dtype) and none for
 * x / x_multi.  Replaces what autograd records for builder.py:107-137.
 *   tp_forward_train : tp_forward + the fp16 pre-GELU activations, into a caller-owned `train_workspace`
 *                      (tp_train_workspace_bytes) that must stay untouched until tp_backward has run.
 *   tp_backward      : dy [B, M, D] contiguous (desc.dtype) -> grads; `raw` = the model's current parameters
 *                      (the same tensors tp_pack_weights packed), `bw_workspace` = scratch
 *                      (tp_backward_workspace_bytes).  Gradients travel in desc.dtype, accumulate in fp32;
 *                      no atomics: results are deterministic. */
typedef struct tp_grads {
    void* q_proj_1_weight;
    void* k_proj_1_0_weight;  void* k_proj_1_0_bias;  void* k_proj_1_2_weight;  void* k_proj_1_2_bias;
    void* v_proj_1_0_weight;  void* v_proj_1_0_bias;  void* v_proj_1_2_weight;  void* v_proj_1_2_bias;
    void* ln_q_1_weight;  void* ln_q_1_bias;  void* ln_k_1_weight;  void* ln_k_1_bias;
    void* ln_v_1_weight;  void* ln_v_1_bias;
    void* clip_attn_in_proj_weight;   void* clip_attn_in_proj_bias;
    void* clip_attn_out_proj_weight;  void* clip_attn_out_proj_bias;
    void* mlp_0_weight;  void* mlp_0_bias;  void* mlp_2_weight;  void* mlp_2_bias;
} tp_grads;
size_t tp_train_workspace_bytes(const tp_desc* desc);
size_t tp_backward_workspace_bytes(const tp_desc* desc);
int tp_forward_train(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
                     const int64_t xm_strides[3], const void* packed_weights, void* out, void* train_workspace,
                     size_t workspace_bytes, void* stream);
int tp_backward(const tp_desc* desc, const void* x_multi, const int64_t xm_strides[3], const tp_weights* raw,
                const void* packed_weights, const void* train_workspace, const void* dy, const tp_grads* grads,
                void* bw_workspace, size_t bw_workspace_bytes, void* stream);
/* the same two with x_multi as its four sources (see tp_forward_parts) */
int tp_forward_train_parts(const tp_desc* desc, const void* x, const int64_t x_strides[3],
                           const void* const xm_parts[4], const int64_t part_strides[3], const void* packed_weights,
                           void* out, void* train_workspace, size_t workspace_bytes, void* stream);
int tp_backward_parts(const tp_desc* desc, const void* const xm_parts[4], const int64_t part_strides[3],
                      const tp_weights* raw, const void* packed_weights, const void* train_workspace, const void* dy,
                      const tp_grads* grads, void* bw_workspace, size_t bw_workspace_bytes, void* stream);

/* The backward's weight-gradient contraction on its own (what autograd computes for every nn.Linear of
 * builder.py:59-83):  dw[n_out, k_in] = sum over the `rows` token rows of dy[r, n] * x[r, k], read from the ROW-MAJOR
 * activations — no transposed copies.  dy [rows, n_out] and x [rows, k_in] are `dtype` with row strides ldy / ldx
 * (elements, multiples of 8); x may be batch-strided like x_multi (x_rows_per_batch a multiple of 64 and >= 128, or 0).
 * dw is `out_dtype` (TP_BF16 / TP_F16 / TP_F32), contiguous; fp32 accumulation, deterministic.
 * Needs k_in % 256 == 0 and n_out % 8 == 0.  workspace: tp_wgrad_workspace_bytes(n_out, k_in) bytes, 256-byte aligned.
 * With TP_WGRAD_X_TRANSPOSED the activation operand is given transposed instead, x = X^T [k_in, ldx] with ldx a
 * multiple of 1024 and zeros in columns rows .. ldx - 1 (an operand a cast or LayerNorm pass rewrote anyway);
 * dy is still read in place. */
enum { TP_WGRAD_X_TRANSPOSED = 1 };
size_t tp_wgrad_workspace_bytes(int n_out, int k_in);
int tp_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, int x_rows_per_batch, int64_t x_batch_stride,
             int64_t rows, int n_out, int k_in, int dtype, void* dw, int out_dtype, int flags, void* workspace,
             size_t workspace_bytes, void* stream);

/* ---- TokenPacker-HD token assembly (the step right after the projector) --------------------------------
 * Replaces the Python loop + torch.cat of `prepare_inputs_labels_for_multimodal` in mode 'slice'
 * (llava_arch.py:140-154): per image, the h_block x w_block crops in row-major order, the ',' embedding after
 * every crop that is not the last of its row, the '\n' embedding after every row, then (more than one crop)
 * the global-view crop and '\n'.  tokens [n_crops, M, D], sep / ret [D], out [out_rows, D]; all of `dtype`.
 * Image i reads crops first_crop .. and writes rows out_row .. out_row + tp_hd_rows(h, w, M).  Images are listed in
 * order; their crop and row ranges must not overlap and must lie inside `tokens` / `out` (validated).  Rows of `out`
 * between two images are left untouched, so `out` can be the `inputs_embeds` buffer itself with the text embeddings
 * already in place (llava_arch.py:172-191) — the visual tokens then land where the LLM reads them.
 * `crop_map` (device int32[n_crops], or NULL): logical crop c lives in row block crop_map[c] of `tokens` — lets the
 * kernel read the b_max-strided buffer of a RAGGED all-gather in place (tokenpacker_amd.shard.GatheredTokens); the
 * caller guarantees the mapped blocks exist (values are not readable from the host side). */
typedef struct tp_hd_image {
    int32_t first_crop;
    int32_t h_block;
    int32_t w_block;
    int32_t reserved;
    int64_t out_row;
} tp_hd_image;
int64_t tp_hd_rows(int h_block, int w_block, int M);
int tp_hd_assemble(const tp_hd_image* plan, int n_images, const void* tokens, int64_t n_crops, const int32_t* crop_map,
                   const void* sep, const void* ret, void* out, int64_t out_rows, int M, int D, int dtype, void* stream);

/* ---- TokenPacker-HD image slicing (the step before the CLIP tower) ---------------------------------------------
 * Replaces the resize / zero-pad / tile code of the data loader and the eval drivers (llava/train/train.py:695-731):
 * image [3, H, W] fp32 (already normalised) -> crops [h_block*w_block (+1), 3, block, block] fp32: the image resized
 * (F.interpolate bilinear, align_corners = False) to (h_res, w_res) in the top-left of a zero canvas of
 * h_block x w_block blocks, cut row-major; with more than one crop, the canvas resized to (hg, wg) and
 * zero-padded to block x block as the last crop.  The sizes follow the reference's rounding rules
 * (tokenpacker_amd.hd.slice_plan). */
int tp_hd_slice(const float* image, int H, int W, int h_block, int w_block, int h_res, int w_res, int hg, int wg,
                float* crops, int block, void* stream);

/* ---- debug: fp16 saturation scan ---------------------------------------------------------------------------------
 * Every activation between the kernels of tp_forward is fp16 and every epilogue CLAMPS to +-65504 instead of producing
 * inf.  After a tp_forward on `stream` with the same desc / workspace, this scans the nine intermediate
 * buffers — q0, Hkv, H2, KV (qt | u under the absorbed schedule), Q1pre, Q, O, A1, A2 in this order — and writes into counts[i] (device int32[9]) the
 * number of elements at the clamp bound (or NaN).  All zeros = no activation of that forward left the fp16 range. */
#define TP_NUM_DEBUG_BUFFERS 9
int tp_debug_count_saturated(const tp_desc* desc, const void* workspace, size_t workspace_bytes, int32_t* counts,
                             void* stream);

/* The first TP_WORKSPACE_STATUS_BYTES of tp_backward's workspace are a status block like the forward workspace's: int32 word 0 holds
 * the sticky saturation bits of THAT backward (zeroed by the call itself): bit 0 — the incoming dy was not finite, or a GEMM epilogue of
 * the fp16 gradient chain clamped a value to +-65504; bit 1 — the LayerNorm backward did; bit 2 — the attention backward did.  Non-zero
 * means the parameter gradients of that call are not trustworthy (TP_TUNE_BWD_CHAIN = 1 carries the gradients in bf16 instead). */

/* ---- diagnostics counters of this process (read-only; ABI 5: the round-3/4 test hooks tp_test_side_cache_size /
 * tp_test_pair_launch_count under one entry — the other, stateless test hooks and the timing-probe instantiations of the GEMM
 * kernels are NOT in this library: include/tokenpacker_test.h, libtokenpacker_exp.so):
 *   TP_COUNTER_SIDE_STREAMS  entries of the per-caller-stream side-stream cache (tp_release_stream, LRU eviction)
 *   TP_COUNTER_PAIR_LAUNCHES launches of the pair GEMM kernel (tp_gemm_pair.hip) issued so far
 * -1: unknown counter. */
enum { TP_COUNTER_SIDE_STREAMS = 0, TP_COUNTER_PAIR_LAUNCHES = 1 };
long long tp_debug_counter(int which);

/* ---- tuning knobs (benchmarks / deployment policy; defaults are what tp_forward ships with) --------------------
 * Two places hold them (ABI 4):
 *   - a TUNING CONTEXT (tp_tuning_create): a private copy of the table.  Every entry point that takes a tp_desc reads its
 *     knobs from desc->tuning for the whole call — plan, workspace layout, every launch — whatever other threads do to
 *     other contexts or to the process-wide table meanwhile.  This is what a serving worker uses (the reference runs
 *     `generate` on a thread per request, llava/serve/model_worker.py:174): one context per model instance, set once.
 *   - the PROCESS-WIDE table (tp_set_tuning / tp_get_tuning): what a descriptor with tuning == NULL, the per-kernel entry
 *     points (tp_linear ...) and tp_tuning_create's initial copy read.  One array of atomics, NOT scoped to a stream, a
 *     call or a thread: benchmarks and tests flip it between calls; concurrent forwards that rely on it see each
 *     other's changes (several knobs change the summation order, i.e. low bits).
 * Everything else in this header is reentrant (state is per call, per thread (error string) or per caller stream (the
 * forked query-side stream; at most 64 distinct caller streams per device get one, later ones run the query side on the
 * caller's stream)). */
enum { TP_TUNE_GEMM_TILE = 0,   /* 0 auto (full tiles, half-tile tail, all half tiles or all 192-row tiles by CU rounds) | 128 | 256 |
                                   2: every tile of the ping-pong kernel a 128 x 256 half tile | 3: a 192 x 256 tile (plain launches) |
                                   4: auto without 192-row tiles (tests, A/Bs; same bits whatever the tile) */
       TP_TUNE_XCD_SWIZZLE = 1, /* 1 (default) | 0 | 2 = A/B: W-half-resident blocking of an XCD's tiles (measured null) */
       TP_TUNE_FOLD_OUT_PROJ = 2, /* out_proj folded into mlp[0] (W = Wm0·Wout, one GEMM less): 0 (default) auto = on the absorbed
                                     schedule and on the scale_factor-2 schedule with attention in the in-projection epilogues
                                     (TP_TUNE_FUSE_ATTN 0) | 1 always | 2 never; read at PACK time (tp_pack_weights builds the
                                     folded weight only then) and at forward time */
       TP_TUNE_DYNAMIC_TILES = 3, /* 1 (default): persistent GEMMs draw tiles from per-XCD queues | 0: static striding */
       TP_TUNE_Q_SIDE_STREAM = 4, /* 1 (default): the query side runs on a forked side stream | 0: one stream */
       TP_TUNE_RESERVE_CUS = 5,   /* r in 0 .. CUs/8 - 1 (default 0): persistent GEMMs launch (CUs/8 - r) workgroups per XCD, leaving
                                     r CUs per XCD to kernels of other streams (RCCL's all-gather overlapping the next forward) */
       TP_TUNE_ABSORB_KV = 6,     /* K/V in-projection absorbed into the query side (see tp_forward): 0 auto (scale_factor >= 3),
                                     1 never, 2 always */
       TP_TUNE_FUSE_KV_LN = 7,    /* 1 (default): inference: the layer in front of every LayerNorm is computed for its row statistics
                                     only and the in-projection behind it reads that layer's INPUT through a pre-multiplied weight
                                     (K/V side: Wc = W'·W2, no H2 written — on the plain schedule through the in-projection GEMM, on
                                     the absorbed schedule through qt and the per-head V GEMM, the attention kernel walking Hkv; query
                                     side: W'q·Wq1, no Q1pre written) | 0: the pre-LayerNorm activations are written and read back.
                                     Read at PACK time (the pre-multiplied weights are built only then) and at forward time */
       TP_TUNE_LN_MERGE = 8,      /* 0 (default): inference, a LayerNorm's consumer on the 128-tile kernel (small batches) merges the
                                     producer's (mean, M2) slabs itself — no ln_finalize launch | 1: always the separate launch */
       TP_TUNE_FUSE_ATTN = 9,    /* inference, scale_factor 2, fused LayerNorm chain, no attn_mask: 0 (default) the first K/V layer reads
                                     the tower's rows in REGION-MAJOR order (a region's 4 tokens = 4 consecutive rows of every K/V-side
                                     tensor) and region attention runs inside the epilogues of the K and V in-projection GEMMs — K, V
                                     are never written, no attention kernel | 1 off (raster rows, tp_region_attention's kernel) |
                                     2 region-major rows + the separate attention kernel (A/B, tests) */
       TP_TUNE_SPLIT_K = 10,      /* small batches (B <= 3 at the shipped shapes), inference: the two K = 4096 GEMMs of the path (first
                                     K/V layer, mlp[2]) — 80 and 64 workgroups walking 64 K-slabs each at B = 1, a serial chain of
                                     ~0.65 us steps — are split over K into up to 8 groups of the 128-tile kernel with fp32 partials
                                     and a fixed-order reduction kernel that applies the epilogue (B = 1: 0.157 -> 0.137 ms).
                                     0 (default since round 3) / 1: on where at least 4 K-groups fit | 2: off.  Deterministic, but NOT
                                     the summation order of the unsplit kernels: with 0 / 1 the low bits of a batch of <= 3 images
                                     differ from the same images inside a larger batch; 2 keeps an image's bits independent of the
                                     batch it travels in (what rounds 1-2 shipped; the reference's eager PyTorch does not have that
                                     property either). */
       TP_TUNE_SMALL_GEMM_WAVES = 11, /* the 128 x 128-tile kernel as 4 waves of 64 x 64 or 8 waves of 32 x 64 (bit-identical): 0 (default)
                                     auto by the number of workgroups of the launch | 4 | 8.  A launch of the 8-wave form with at
                                     most one workgroup per CU keeps FOUR K-slabs in its LDS ring (three in flight, counted vmcnt;
                                     round 3: a one-image forward 0.132 -> 0.128 ms) | 9: 8 waves with the double buffer (A/B) */
       TP_TUNE_TRI_STATS = 12,    /* inference, fused LayerNorm chain: 0 (default, round 3) the layer in front of a LayerNorm is replaced, for
                                     its statistics, by the UPPER-TRIANGULAR factor R of its centred weight (W2c = Q R, Householder QR at
                                     pack time): var = ||R h + c~||^2 / E, a sum of squares, on 40 of the 64 (N-tile, K-tile) pairs; the
                                     consumers use centred chain weights (W'·W2c, W'·b2c) and need no mean at all | 1: the full
                                     statistics GEMM on W2 with (mean, M2), as rounds 1-2 shipped */
       TP_TUNE_PAIR_GEMM = 13,    /* the 256 x 128-tile "pair" kernel (tp_gemm_pair.hip: two co-resident 4-wave workgroups per CU, a tile's
                                     epilogue under the other workgroup's MFMAs; bit-identical to the other GEMM kernels): 0 (default) for
                                     the short-K launches of 1.5 .. 2.5 rounds of its 512 workgroups, where it measured faster — on
                                     full-chip launches its 1.5 x operand traffic loses 12-17 % | 1 never | 2 wherever it is supported */
       TP_TUNE_PAIR_STAGGER = 14, /* percent (default 100) of half a tile period by which the second workgroup of each CU starts late in
                                     a pair-kernel launch (0: both start together — their epilogues then coincide for ever) */
       TP_TUNE_PAIR_DEBUG = 15,   /* libtokenpacker_exp.so ONLY (`make exp`; this library refuses a non-zero value: the probe instantiations are
                                     not built into it).  Timing probes of the pair kernel (fp16 -> fp16 plain launches; results are GARBAGE with 1..4): low 3 bits 1 no
                                     DMA in the K loop | 2 no fragment reads | 3 no barriers | 4 no MFMAs; + 8: one workgroup per CU;
                                     bits 4.. (value >> 4): the same kind of probe of the ping-pong kernel's K loop (tools/loop_probe.py:
                                     1 / 3 / 15 no b0 / W / any fragment reads, 16 no DMA, 31 neither, 64 no MFMAs, 79, 80) */
       TP_TUNE_DECOUPLE_K = 16,   /* scale_factor 2, attention in the in-projection epilogues, centred chain (all defaults): 0 (default) the K
                                     launch behind the statistics, LayerNorm fold in its epilogue (the round-5 form) | 1 the K launch writes
                                     RAW logits Q·(Hkv·Wcc^T + dcc) — the K rows' rstd is applied by the V launch, the K bias is
                                     softmax-invariant — so it does not wait for the row statistics and runs on the side stream BESIDE the
                                     statistics launch | 2 raw logits on the caller's stream (same bits as 1: the A/B of the placement
                                     alone).  Measured null at B = 32 .. 256 (profiles/r06b_decouple_k_ab.txt): kept as the record of
                                     that A/B, parity-tested (tests/test_gpu_round6.py) */
       TP_TUNE_BWD_CHAIN = 17,    /* tp_backward of a bf16 model: 0 (default) gradients travel between the backward's kernels in FP16 behind a
                                     dynamic power-of-two scale (S = 2^k brings amax(dy) to (16, 32], computed on the device per call; every
                                     parameter gradient is multiplied by 1 / S — exact — when its fp32 sum is cast to bf16): the weight gradients
                                     read the forward's saved fp16 activations in place instead of through a bf16 copy each (6 of 9 cast passes
                                     gone), and the chain keeps 11 mantissa bits instead of 8 | 1 gradients travel in bf16 (rounds 1-5).  fp16
                                     models are not affected (their gradients travel in fp16, unscaled, as before) */
       TP_TUNE_COUNT_ = 18 };
int tp_set_tuning(int key, int value);
int tp_get_tuning(int key);                  /* the library's current value (not a binding's shadow copy); -1: bad key */
typedef struct tp_tuning tp_tuning;          /* opaque; host memory owned by the library until tp_tuning_destroy */
tp_tuning* tp_tuning_create(void);           /* a copy of the process-wide table of the moment; NULL: out of memory */
void tp_tuning_destroy(tp_tuning* t);        /* (no call that was handed `t` may still be running) */
int tp_tuning_set(tp_tuning* t, int key, int value);
int tp_tuning_get(const tp_tuning* t, int key);          /* -1: bad key / NULL */

/* Per-caller-stream state: tp_forward forks the query side onto a side stream it keeps per (device, caller stream).
 * A caller that destroys a stream it has passed to tp_forward releases that state first — a later stream the runtime
 * hands out at the same address would otherwise inherit it.  Synchronises the side stream; a stream the library has
 * never seen is not an error.  The cache holds 64 entries per process; beyond that the least recently used one is
 * released the same way. */
int tp_release_stream(void* stream);

/* ---- CU-free all-gather of the projected tokens ("one all-gather ... before the LLM") ----------------
 * One process per GPU.  Every rank's projector writes its shard into rows [lo, hi) of a receive buffer [total, M, D]
 * (tp_forward's `out`), and the shard travels to the same rows of every peer's buffer by ONE hipMemcpyAsync per peer —
 * SDMA engines over that peer's xGMI link, no compute unit — instead of RCCL's all-gather kernels, which need the CUs the
 * next forward's persistent GEMMs own.  The reference has no counterpart (it never calls torch.distributed); this is the
 * data-path step the north_star prescribes.  tokenpacker_amd/shard.py (DirectGather) drives it; INTEGRATION.md §4 shows
 * the protocol.  Everything below only enqueues; buffers, flags, streams and events are the caller's.
 *
 *   tp_gather_export  IPC handle of the ALLOCATION `ptr` lives in + ptr's offset inside it (hipMemGetAddressRange +
 *                     hipIpcGetMemHandle): what a peer process needs to address `ptr`.
 *   tp_gather_open    maps a peer's allocation into this process (peer access enabled lazily); tp_gather_close unmaps it.
 *   tp_gather_push    for each of n_peers: copy `bytes` from `src` to dst[j], then the 4-byte sequence number in
 *                     *seq_cell to dst_flag[j], both on streams[j] (stream order: the flag cannot land before the data).
 *                     use_cus = 0: hipMemcpyDeviceToDeviceNoCU (SDMA) | 1: the runtime's choice (A/B).
 *   tp_gather_sync    ONE-wave kernel on `stream`: waits until flags[p] has reached wait_seq for every p != rank
 *                     (wrap-safe signed distance), then stores publish_seq to *seq_cell (NULL: nothing).  A wait longer than
 *                     timeout_ms (<= 0: 30 s) sets bit 0 of *status (device int32, may be NULL) and gives up — the caller
 *                     reads status when it synchronises.  flags: device uint32[world] in THIS rank's memory, zeroed
 *                     once, slot p written only by rank p's tp_gather_push.
 *   tp_gather_alloc_flags  the flag array as FINE-GRAINED device memory (hipExtMallocWithFlags), zeroed: it is written by
 *                     other devices' copy engines while a kernel of this device polls it, and ordinary device memory is only
 *                     coherent across agents at kernel boundaries.  The one allocation the library performs (on request, a few
 *                     hundred bytes; tp_gather_free_flags releases it) — the caller's framework allocator cannot provide it. */
#define TP_IPC_HANDLE_BYTES 64
int tp_gather_alloc_flags(void** flags, size_t bytes);
int tp_gather_free_flags(void* flags);
int tp_gather_export(const void* ptr, void* handle /* [TP_IPC_HANDLE_BYTES] */, uint64_t* offset);
int tp_gather_open(const void* handle, void** base);
int tp_gather_close(void* base);
int tp_gather_push(int n_peers, void* const* dst, const void* src, size_t bytes, void* const* dst_flag,
                   const uint32_t* seq_cell, void* const* streams, int use_cus);
int tp_gather_sync(const uint32_t* flags, int world, int rank, uint32_t wait_seq, uint32_t* seq_cell,
                   uint32_t publish_seq, int32_t* status, int timeout_ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TOKENPACKER_H_ */
