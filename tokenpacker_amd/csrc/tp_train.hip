// tp_train.hip — training entry points of libtokenpacker_hip.so: the forward that keeps what the backward needs,
// and the backward pass of the projector (gradients of all 23 parameters; the CLIP features come from a frozen
// tower and get none — reference llava/train/train.py:950-953, clip_encoder.py:46 `@torch.no_grad()`).
//
// Backward schedule (reverse of tp_api.hip's forward; R = B*576 fine tokens, Rq = B*M coarse tokens, G = the
// model dtype in which gradients travel):
//   mlp[2]      dW = dy^T·A2, db = colsum(dy);         dZ2 = (dy·Wm2) * gelu'(Z2)
//   mlp[0]      dW = dZ2^T·A1, db = colsum(dZ2);       dA1 = dZ2·Wm0
//   out_proj    dW = dA1^T·O,  db = colsum(dA1);       dO  = dA1·Wout
//   attention   (Q, K, V, dO) -> dQ, dK, dV                                   region_attention_bwd_kernel
//   in_proj     dW{q,k,v} = d{Q,K,V}^T·LN(.), db = colsum;   d(LN out) = d{Q,K,V}·W{q,k,v}
//   LayerNorms  d(pre-LN), dgamma, dbeta                                      ln_backward_kernel
//   q_proj_1    dW = dQ1pre^T·q0
//   k/v_proj[2] dW = dH2^T·Hkv, db = colsum(dH2);      dZ1 = (dH2·W2) * gelu'(Z1)
//   k/v_proj[0] dW = dZ1^T·x_multi, db = colsum(dZ1)
// Every contraction runs on the forward's MFMA kernels: dgrad as linear(dY, W^T), wgrad with dY (and, wherever its dtype allows,
// the activation) read IN PLACE as K-major operands, split over the token dimension into fp32 partials that a reduction kernel
// sums and casts.
// Round 6 (the non-GEMM work): (1) a bf16 model's gradients travel in FP16 behind a dynamic power-of-two scale
// (bw_f16_chain below) — the saved fp16 activations are then weight-gradient operands as they lie (six cast / transpose passes
// gone) and the chain keeps 11 mantissa bits; (2) the LayerNorm backward also writes the LayerNorm's OUTPUT row-major (the
// in-projection's weight gradient reads it in place: the three normalising transposes — 0.82 ms per B = 256 step — are gone) and
// the column sums of its dx (the bias gradient of the layer in front); (3) dy is read once for mlp[2]'s bias gradient and for
// max |dy|; (4) in_proj_bias' v third is colsum(dO) (= colsum(dV): every head's softmax sums to 1), a 4 x smaller pass.
// B = 256: 15.5-15.9 -> 14.5 ms per step incl. optimizer; B = 32: 3.05 -> 2.90 (profiles/r06j_*, r06l_*, r06n_*).
#include "tp_internal.h"

namespace tp {

static inline size_t up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
static inline int pad_tokens(long long r) { return (int)((r + 1023) / 1024 * 1024); }   // any split in {1,..,16} keeps K % 64 == 0

struct BwLayout {
    // transposed weights (G) for the dgrad GEMMs, fp32 LayerNorm affines
    size_t wt_m2, wt_m0, wt_out, wt_in, wt_2;        // wt_in: [3][E,E] (q,k,v);  wt_2: [2][E,E] (k,v)
    size_t ln_g, ln_b;                                // [3][E] fp32 each (q,k,v)
    // coarse-token side
    size_t dz2, da1, dO, dQ, dq1, dQ1pre;
    // fine-token side ([2] = k, v)
    size_t dKV, dkv1, dH2, dZ1;
    // ONE transposed-operand scratch: every weight gradient whose activation operand must be rewritten (fp16 activation of a
    // bf16 model cast, LayerNorm input normalised) transposes it here and consumes it in the next launch — never two at once
    size_t xt;
    // fallback paths only (hidden_size not a multiple of 256 / a grid whose images are not whole 64-row K-tiles): transposed
    // copies of dy, dZ2, dZ1 and x_multi; kNoSlab when the in-place (K-major operand) paths serve the shape
    size_t dyT, dz2T, dZ1T, xmT;
    size_t counters;                                  // tile-queue heads of the persistent GEMM launches (zeroed once)
    size_t part, colpart, lnpart;                     // fp32 partials: split-K wgrad, column sums, LN affine grads
    size_t redscratch;                                // stage-1 output of the many-parts reduction
    // fp16 chain of a bf16 model (TP_TUNE_BWD_CHAIN = 0): dy scaled by a dynamic power of two and cast [Rq, D] f16, the scale
    // pair (S, 1 / S) on the device, 1024 amax partials; kNoSlab otherwise
    size_t dy16, scale, amaxpart;
    size_t status;                                    // 256 B at offset 0: int32[0] = sticky saturation bits of the fp16 gradient chain (tp_backward_status_bytes)
    size_t part_bytes;
    size_t total;
    int Rp, Rqp;
};

// Whether the first-layer weight gradient reads x_multi in place (wgrad_tt_supported's shape conditions; strides are
// multiples of 8 by tp_forward's own argument check): images must be whole 64-row K-tiles.
static bool xm_inplace(int grid) { const int N = grid * grid; return N % 64 == 0 && N >= 128; }

// Whether the backward of this descriptor carries its gradients in fp16 behind a dynamic power-of-two scale (a bf16 model; the
// default) instead of in the model dtype.  Why: the forward saves its activations in fp16, and an MFMA takes one
// operand type — with bf16 gradients every weight gradient first re-wrote its activation operand as bf16 (nine cast / transpose
// passes, 1.3 ms per B = 256 step); with fp16 gradients six of the nine read the saved activation in place, and the gradients
// keep 11 mantissa bits instead of 8 through the chain.  fp16's range is handled like loss scaling handles it, but per call and
// on the device: S = 2^k brings amax(dy) to (16, 32], everything downstream is linear in dy, every parameter gradient is
// multiplied by 1 / S (exact) when its fp32 sum is cast to the model dtype.
static bool bw_f16_chain(const tp_desc* d) { return d->dtype == TP_BF16 && tuning(TP_TUNE_BWD_CHAIN) == 0; }

static BwLayout bw_layout(int B, int grid, int s, int D, bool f16_chain) {
    BwLayout L{};
    const size_t N = (size_t)grid * grid, G = grid / s, M = G * G, E = kEmbed;
    const size_t R = (size_t)B * N, Rq = (size_t)B * M;
    L.Rp = pad_tokens((long long)R); L.Rqp = pad_tokens((long long)Rq);
    const size_t Rp = L.Rp, Rqp = L.Rqp;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up(off + bytes); return o; };
    auto take_if = [&](bool need, size_t bytes) { return need ? take(bytes) : kNoSlab; };
    L.status = take(256);                             // (first: the caller finds it at offset 0, like the forward workspace's status block)
    L.wt_m2 = take((size_t)D * D * 2); L.wt_m0 = take(E * D * 2); L.wt_out = take(E * E * 2);
    L.wt_in = take(3 * E * E * 2); L.wt_2 = take(2 * E * E * 2);
    L.ln_g = take(3 * E * 4); L.ln_b = take(3 * E * 4);
    L.dz2 = take(Rq * D * 2); L.da1 = take(Rq * E * 2);
    L.dO = take(Rq * E * 2); L.dQ = take(Rq * E * 2);
    L.dq1 = take(Rq * E * 2); L.dQ1pre = take(Rq * E * 2);
    L.dKV = take(2 * R * E * 2); L.dkv1 = take(2 * R * E * 2);
    L.dH2 = take(2 * R * E * 2);
    L.dZ1 = take(R * 2 * E * 2);
    {
        const size_t a = (size_t)D * Rqp * 2, b = E * Rp * 2;        // the largest operands: A2^T [D, Rq], H2^T / Hkv^T [E, R]
        L.xt = take(a > b ? a : b);
    }
    const bool inplace_d = D % 256 == 0, inplace_xm = xm_inplace(grid);
    L.dyT = take_if(!inplace_d, D * Rqp * 2); L.dz2T = take_if(!inplace_d, D * Rqp * 2);
    L.dZ1T = take_if(!inplace_xm, 2 * E * Rp * 2); L.xmT = take_if(!inplace_xm, (size_t)kMulti * Rp * 2);
    // split-K partials: S splits of an [Nout, Kin] weight with S * tiles(256^2) <= ~512
    size_t wmax = (size_t)D * D;
    if ((size_t)2 * E * kMulti > wmax) wmax = (size_t)2 * E * kMulti;
    L.part_bytes = 2 * wmax * 4;
    if (L.part_bytes < (size_t)16 * D * E * 4) L.part_bytes = (size_t)16 * D * E * 4;
    L.part = take(L.part_bytes);
    const size_t cmax = (size_t)D > 2 * E ? (size_t)D : 2 * E;
    L.colpart = take((Rp / 64) * cmax * 4);
    L.lnpart = take((size_t)256 * 3 * E * 4);
    L.redscratch = take((size_t)kReduceSlices * cmax * 4);
    L.counters = take(64 * 32 * 4);
    L.dy16 = take_if(f16_chain, Rq * D * 2);
    L.scale = take_if(f16_chain, 256);
    L.amaxpart = take_if(f16_chain, (size_t)kColsumAmaxParts * 4);
    L.total = off;
    return L;
}

#define TP_TRY(expr) do { int rc_ = (expr); if (rc_ != TP_OK) return rc_; } while (0)

// ---- weight gradients from K-major operands ---------------------------------------------------------------------
static int wgrad_splits(long long R, int Nout, int Kin) {
    const long long tiles = (long long)((Nout + 255) / 256) * (Kin / 256);
    int S = 1;
    while (S < 16 && tiles * S < 256 && R / (S * 2) >= 512) S *= 2;
    return S;
}
size_t wgrad_tt_part_bytes(int Nout, int Kin) { return (size_t)16 * Nout * Kin * 4; }

bool wgrad_tt_supported(long long R, int Nout, int Kin, const WgradX& X, long long ldy) {
    if (R <= 0 || Nout <= 0 || Kin <= 0 || Kin % 256 != 0 || Nout % 8 != 0 || ldy % 8 != 0 || X.ldx % 8 != 0) return false;
    if (X.rows_per_batch > 0 && X.rows_per_batch < R &&
        (X.rows_per_batch % 64 != 0 || X.rows_per_batch < 128 || X.batch_stride % 8 != 0)) return false;
    if (X.parts && (X.n_part <= 0 || X.n_part % 256 != 0 || Kin != 4 * X.n_part)) return false;
    return true;
}

int wgrad_tt_launch(int dtype, const void* dY, long long ldy, const WgradX& X, long long R, int Nout, int Kin,
                    float* part, size_t part_bytes, int out_dtype, void* grad_out, int* counters, hipStream_t stream,
                    void* grad_out_hi, int split_row, const float* out_scale) {
    if (!wgrad_tt_supported(R, Nout, Kin, X, ldy)) {
        set_error("tp wgrad: unsupported shape R=%lld Nout=%d Kin=%d (need Kin %% 256 == 0, strides %% 8 == 0, batches of a multiple of 64 rows)",
                  R, Nout, Kin);
        return TP_ERR_INVALID_ARG;
    }
    const int S = wgrad_splits(R, Nout, Kin);
    if ((size_t)S * Nout * Kin * 4 > part_bytes) { set_error("tp wgrad: split partial buffer too small"); return TP_ERR_WORKSPACE; }
    const long long Ks = ((R + S - 1) / S + 63) / 64 * 64;          // contraction rows per split, whole K-tiles
    GemmArgs a{};
    a.A = (const char*)dY; a.lda_bytes = ldy * 2; a.a_gs = Ks * ldy * 2;
    a.W = X.parts ? (const char*)X.parts[0] : (const char*)X.x; a.ldw_bytes = X.ldx * 2; a.w_gs = 0;
    if (X.parts) { for (int i = 0; i < 4; ++i) a.W_parts[i] = (const char*)X.parts[i]; a.n_part = X.n_part; }
    if (X.rows_per_batch > 0 && X.rows_per_batch < R) {
        a.tt_tpb = X.rows_per_batch / 64;                // K-tiles per batch (>= 2), divided by with a 32-bit magic number
        a.tt_bmagic = (unsigned)(((1ull << 32) + a.tt_tpb - 1) / a.tt_tpb);
        a.w_batch_stride_bytes = X.batch_stride * 2;
    }
    a.C = (char*)part; a.ldc = Kin; a.c_gs = (long long)Nout * Kin * 4;
    a.M = Nout; a.N = Kin; a.K = (int)Ks; a.groups = S; a.tt_rows = R; a.rows_per_batch = Nout; a.tile = 256;
    a.tile_counters = S <= 4 ? counters : nullptr;       // (a queue slot holds [4 groups][8 XCDs] heads)
    TP_TRY(gemm_launch(dtype, TP_F32, a, stream));
    if (!grad_out_hi || split_row <= 0 || split_row >= Nout)
        return bw_reduce_parts_launch(out_dtype, part, (long long)Nout * Kin, S, (long long)Nout * Kin, grad_out, stream, out_scale);
    TP_TRY(bw_reduce_parts_launch(out_dtype, part, (long long)Nout * Kin, S, (long long)split_row * Kin, grad_out, stream, out_scale));
    return bw_reduce_parts_launch(out_dtype, part + (size_t)split_row * Kin, (long long)Nout * Kin, S,
                                  (long long)(Nout - split_row) * Kin, grad_out_hi, stream, out_scale);
}

// dW[Nout, Kin] = dY^T · X with dY [R, Nout] read in place (K-major) and X given TRANSPOSED, XT [Kin, rpad] (zero beyond
// column R; rpad a multiple of 1024): the operand a cast / LayerNorm pass had to rewrite anyway.
int wgrad_tn_launch(int dtype, const void* dY, long long ldy, const void* XT, long long rpad, long long R, int Nout, int Kin,
                    float* part, size_t part_bytes, int out_dtype, void* grad_out, int* counters, hipStream_t stream,
                    const float* out_scale) {
    if (R <= 0 || Nout <= 0 || Kin % 256 != 0 || Nout % 8 != 0 || ldy % 8 != 0 || rpad % 1024 != 0 || rpad < R) {
        set_error("tp wgrad: unsupported shape R=%lld rpad=%lld Nout=%d Kin=%d", R, rpad, Nout, Kin);
        return TP_ERR_INVALID_ARG;
    }
    const long long tiles = (long long)((Nout + 255) / 256) * (Kin / 256);
    int S = 1;
    while (S < 16 && tiles * S < 256 && (rpad / (S * 2)) % 64 == 0 && rpad / (S * 2) >= 256) S *= 2;
    if ((size_t)S * Nout * Kin * 4 > part_bytes) { set_error("tp wgrad: split partial buffer too small"); return TP_ERR_WORKSPACE; }
    const long long Ks = rpad / S;
    GemmArgs a{};
    a.A = (const char*)dY; a.lda_bytes = ldy * 2; a.a_gs = Ks * ldy * 2;
    a.W = (const char*)XT; a.ldw_bytes = rpad * 2; a.w_gs = Ks * 2;
    a.C = (char*)part; a.ldc = Kin; a.c_gs = (long long)Nout * Kin * 4;
    a.M = Nout; a.N = Kin; a.K = (int)Ks; a.groups = S; a.tt_rows = R; a.tt_w_kcontig = 1; a.rows_per_batch = Nout; a.tile = 256;
    a.tile_counters = S <= 4 ? counters : nullptr;
    TP_TRY(gemm_launch(dtype, TP_F32, a, stream));
    return bw_reduce_parts_launch(out_dtype, part, (long long)Nout * Kin, S, (long long)Nout * Kin, grad_out, stream, out_scale);
}

}  // namespace tp

using namespace tp;


extern "C" {

size_t tp_wgrad_workspace_bytes(int n_out, int k_in) {
    if (n_out <= 0 || k_in <= 0) { set_error("tp_wgrad_workspace_bytes: bad shape %d x %d", n_out, k_in); return 0; }
    return wgrad_tt_part_bytes(n_out, k_in);
}

int tp_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, int x_rows_per_batch, int64_t x_batch_stride,
             int64_t rows, int n_out, int k_in, int dtype, void* dw, int out_dtype, int flags, void* workspace,
             size_t workspace_bytes, void* stream) {
    if (!dy || !x || !dw || !workspace) { set_error("tp_wgrad: NULL argument"); return TP_ERR_INVALID_ARG; }
    if ((dtype != TP_BF16 && dtype != TP_F16) || (out_dtype != TP_BF16 && out_dtype != TP_F16 && out_dtype != TP_F32)) {
        set_error("tp_wgrad: unsupported dtypes in=%d out=%d", dtype, out_dtype);
        return TP_ERR_INVALID_ARG;
    }
    if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dw & 15) || ((uintptr_t)workspace & 255)) {
        set_error("tp_wgrad: dy / x / dw must be 16-byte aligned, the workspace 256-byte aligned");
        return TP_ERR_INVALID_ARG;
    }
    if (flags & ~TP_WGRAD_X_TRANSPOSED) { set_error("tp_wgrad: unknown flags %d", flags); return TP_ERR_INVALID_ARG; }
    if (flags & TP_WGRAD_X_TRANSPOSED)
        return wgrad_tn_launch(dtype, dy, ldy, x, ldx, rows, n_out, k_in, (float*)workspace, workspace_bytes, out_dtype, dw,
                               nullptr, (hipStream_t)stream);
    WgradX X{x, ldx, x_rows_per_batch, x_batch_stride, nullptr, 0};
    return wgrad_tt_launch(dtype, dy, ldy, X, rows, n_out, k_in, (float*)workspace, workspace_bytes, out_dtype, dw, nullptr,
                           (hipStream_t)stream);
}

size_t tp_train_workspace_bytes(const tp_desc* desc) {
    tp::TuningScope tuning_scope(desc);
    if (validate_desc(desc) != TP_OK) return 0;
    return workspace_layout(desc->batch, desc->raw_grid, desc->scale_factor, desc->hidden_size, true).total;
}

size_t tp_backward_workspace_bytes(const tp_desc* desc) {
    tp::TuningScope tuning_scope(desc);
    if (validate_desc(desc) != TP_OK) return 0;
    return bw_layout(desc->batch, desc->raw_grid, desc->scale_factor, desc->hidden_size, bw_f16_chain(desc)).total;
}

int tp_forward_train(const tp_desc* desc, const void* x, const int64_t x_strides[3], const void* x_multi,
                     const int64_t xm_strides[3], const void* packed_weights, void* out, void* train_workspace,
                     size_t workspace_bytes, void* stream) {
    return forward_impl(desc, x, x_strides, x_multi, xm_strides, packed_weights, out, train_workspace, workspace_bytes,
                        stream, nullptr, true);
}

int tp_forward_train_parts(const tp_desc* desc, const void* x, const int64_t x_strides[3],
                           const void* const xm_parts[4], const int64_t part_strides[3], const void* packed_weights,
                           void* out, void* train_workspace, size_t workspace_bytes, void* stream) {
    if (!xm_parts) { set_error("tp_forward_train_parts: xm_parts is NULL"); return TP_ERR_INVALID_ARG; }
    return forward_impl(desc, x, x_strides, nullptr, part_strides, packed_weights, out, train_workspace, workspace_bytes,
                        stream, nullptr, true, xm_parts);
}

static int backward_impl(const tp_desc* desc, const void* x_multi, const void* const* xm_parts, const int64_t xm_strides[3],
                         const tp_weights* raw, const void* packed_weights, const void* train_workspace, const void* dy,
                         const tp_grads* grads, void* bw_workspace, size_t bw_workspace_bytes, void* stream_) {
    tp::TuningScope tuning_scope(desc);
    TP_TRY(validate_desc(desc));
    if (xm_parts) {
        for (int i = 0; i < 4; ++i) if (!xm_parts[i]) { set_error("tp_backward: x_multi part %d is NULL", i); return TP_ERR_INVALID_ARG; }
        x_multi = xm_parts[0];
    }
    if (!x_multi || !xm_strides || !raw || !packed_weights || !train_workspace || !dy || !grads || !bw_workspace) {
        set_error("tp_backward: NULL argument");
        return TP_ERR_INVALID_ARG;
    }
    {
        const void* const* gp = reinterpret_cast<const void* const*>(grads);
        const void* const* wp = reinterpret_cast<const void* const*>(raw);
        for (size_t i = 0; i < sizeof(tp_grads) / sizeof(void*); ++i)
            if (!gp[i] || !wp[i]) { set_error("tp_backward: weight / gradient pointer #%zu is NULL", i); return TP_ERR_INVALID_ARG; }
    }
    if (desc->out_dtype != desc->dtype) { set_error("tp_backward: out_dtype must equal dtype"); return TP_ERR_INVALID_ARG; }
    const int B = desc->batch, g = desc->raw_grid, s = desc->scale_factor, D = desc->hidden_size;
    // MT: the model dtype — weights, x_multi, dy and the parameter gradients.  GT: the dtype gradients TRAVEL in between the
    // backward's kernels: fp16 behind a dynamic power-of-two scale for a bf16 model (bw_f16_chain), the model dtype otherwise.
    const int MT = desc->dtype;
    const bool f16_chain = bw_f16_chain(desc);
    const int GT = f16_chain ? TP_F16 : MT;
    const int N = g * g, Gq = g / s, M = Gq * Gq, E = kEmbed;
    const int R = B * N, Rq = B * M;
    const WorkspaceLayout W = workspace_layout(B, g, s, D, true);
    const BwLayout L = bw_layout(B, g, s, D, f16_chain);
    if (bw_workspace_bytes < L.total) {
        set_error("tp_backward: workspace %zu B < required %zu B", bw_workspace_bytes, L.total);
        return TP_ERR_WORKSPACE;
    }
    if (((uintptr_t)bw_workspace & 255) || ((uintptr_t)train_workspace & 255) || ((uintptr_t)dy & 15)) {
        set_error("tp_backward: workspaces must be 256-byte aligned, dy 16-byte aligned");
        return TP_ERR_INVALID_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const char* fw = (const char*)train_workspace;
    char* bw = (char*)bw_workspace;
    const int Rp = L.Rp, Rqp = L.Rqp;
    float* part = (float*)(bw + L.part);
    float* colpart = (float*)(bw + L.colpart);
    float* redscratch = (float*)(bw + L.redscratch);
    const long long kvE = (long long)R * E;
    float* const scale = f16_chain ? (float*)(bw + L.scale) : nullptr;          // [0] = S, [1] = 1 / S (device)
    const float* const inv_scale = scale ? scale + 1 : nullptr;

    int* counters = tuning(TP_TUNE_DYNAMIC_TILES) ? (int*)(bw + L.counters) : nullptr;
    if (counters) {
        hipError_t e = hipMemsetAsync(counters, 0, 64 * 32 * 4, stream);
        if (e != hipSuccess) { set_error("tp_backward: hipMemsetAsync: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    // Sticky saturation word of this backward (offset 0 of the workspace; zeroed here): bit 0 — dy itself was not finite / a GEMM
    // epilogue of the fp16 chain clamped, bit 1 — LayerNorm backward, bit 2 — attention backward.  fp16's range is what the dynamic
    // scale buys 2^11 of headroom in; a chain that outgrows it (a LayerNorm row of ~zero variance multiplies by rstd ~ 1e3) clamps
    // instead of producing inf, and says so here — the module warns once and names TP_TUNE_BWD_CHAIN = 1.
    int* const sat = GT == TP_F16 ? (int*)(bw + L.status) : nullptr;
    {
        hipError_t e = hipMemsetAsync(bw + L.status, 0, 256, stream);
        if (e != hipSuccess) { set_error("tp_backward: hipMemsetAsync(status): %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    int launch_no = 0;
    auto launch = [&](int in_dt, int out_dt, GemmArgs& a) -> int {
        a.tile_counters = (counters && launch_no < 64 && a.groups <= 4) ? counters + 32 * launch_no++ : nullptr;
        if (out_dt == TP_F16) { a.sat_flag = sat; a.sat_bit = 1; }
        return gemm_launch(in_dt, out_dt, a, stream);
    };
    // ---- small helpers ------------------------------------------------------------------------------------
    // transpose (+cast) of a contiguous-row matrix; optional LayerNorm application and column sums
    auto T = [&](int sdt, const void* src, long long ld, int rows, int cols, void* dst, int rpad, const float* mr = nullptr,
                 const float* gam = nullptr, const float* bet = nullptr, float* csum = nullptr) -> int {
        return bw_transpose_launch(sdt, GT, src, ld, rows, 0, rows, cols, dst, rpad, rpad, mr, gam, bet, csum, stream);
    };
    // bias gradient from the column-sum partials the last transpose left behind
    auto bias_grad = [&](int rpad, int cols, void* out) -> int {
        return bw_reduce_many_parts_launch(MT, colpart, cols, rpad / 64, cols, out, redscratch, stream, inv_scale);
    };
    // dX[rows, Kin] = dY[rows, Nout] · W[Nout, Kin]  with W^T [Kin, Nout] given
    auto dgrad = [&](const void* dY, long long ldy, int rows, int Nout, const void* WT, int Kin, void* dX, long long ldx,
                     int flags = 0, const void* Z = nullptr, long long ldz = 0, int out_dt = -1) -> int {
        GemmArgs a = plain_gemm(dY, ldy, WT, dX, ldx, rows, Kin, Nout, nullptr, flags);
        a.Z = (const char*)Z; a.ldz = ldz;
        return launch(GT, out_dt < 0 ? GT : out_dt, a);
    };
    // dW[Nout, Kin] = dY^T[Nout, rpad] · X^T[Kin, rpad]^T, split over the token dimension
    auto wgrad = [&](const void* dYT, const void* XT, int Nout, int Kin, int rpad, void* grad_out) -> int {
        const long long tiles = (long long)((Nout + 255) / 256) * ((Kin + 255) / 256);
        int S = 1;
        while (S < 16 && tiles * S < 256 && (rpad / (S * 2)) % 64 == 0) S *= 2;
        if ((size_t)S * Nout * Kin * 4 > L.part_bytes) { set_error("tp_backward: split-K partial buffer too small"); return TP_ERR_WORKSPACE; }
        GemmArgs a = plain_gemm(dYT, rpad, XT, part, Kin, Nout, Kin, rpad / S, nullptr, 0);
        a.ldw_bytes = (long long)rpad * 2;
        a.groups = S; a.a_gs = (long long)(rpad / S) * 2; a.w_gs = (long long)(rpad / S) * 2; a.c_gs = (long long)Nout * Kin * 4;
        a.tile = (Nout % 256 == 0 && Kin % 256 == 0) ? 0 : 128;      // auto: 256-tile persistent kernel once S * tiles fills the chip
        TP_TRY(launch(GT, TP_F32, a));
        return bw_reduce_parts_launch(MT, part, (long long)Nout * Kin, S, (long long)Nout * Kin, grad_out, stream, inv_scale);
    };

    // The same from dY in place (K-major GEMM operand, no dY^T): bias gradient by a column-sum pass over dY, weight
    // gradient with the activation operand either in place as well (`X` row-major in the model dtype) or transposed
    // (`XT` [Kin, rpad]: fp16 activations of a bf16 model are cast, LayerNorm inputs normalised, by that transpose).
    auto next_counters = [&]() -> int* { return (counters && launch_no < 64) ? counters + 32 * launch_no++ : nullptr; };
    // (`dt`: the dtype dY is stored in; `sc`: the 1 / S its values carry, NULL for the unscaled incoming dy)
    auto bias_grad_rows = [&](const void* dY, long long ldy, long long rows, int cols, void* out, int dt, const float* sc) -> int {
        const int slices = bw_colsum_rows_launch(dt, dY, ldy, rows, cols, colpart, stream);
        if (slices < 0) return slices;
        return bw_reduce_many_parts_launch(MT, colpart, cols, slices, cols, out, redscratch, stream, sc);
    };
    auto wgrad_rows = [&](const void* dY, long long ldy, long long rows, int rpad, int Nout, int Kin, const void* X, long long ldx,
                          int x_dtype, char* XT_scratch, void* grad_out, const float* mr = nullptr, const float* gam = nullptr,
                          const float* bet = nullptr) -> int {
        const WgradX XR{X, ldx, 0, 0, nullptr, 0};
        if (x_dtype == GT && !mr && wgrad_tt_supported(rows, Nout, Kin, XR, ldy))
            return wgrad_tt_launch(GT, dY, ldy, XR, rows, Nout, Kin, part, L.part_bytes, MT, grad_out, next_counters(), stream,
                                   nullptr, 0, inv_scale);
        if (!XT_scratch) { set_error("tp_backward: a row-major operand the K-major weight gradient cannot read in place (rows %lld, %d x %d)", rows, Nout, Kin); return TP_ERR_INVALID_ARG; }
        TP_TRY(T(x_dtype, X, ldx, (int)rows, Kin, XT_scratch, rpad, mr, gam, bet));
        return wgrad_tn_launch(GT, dY, ldy, XT_scratch, rpad, rows, Nout, Kin, part, L.part_bytes, MT, grad_out, next_counters(),
                               stream, inv_scale);
    };
    const bool inplace = (D % 256 == 0);                // (E = 1024 always is): every weight's Kin is a multiple of 256

    // ---- operands the backward needs in its own layout --------------------------------------------------------
    // transposed weights (model dtype): W [out, in] -> W^T [in, out]
    {
        TransposeBatch tb{};                             // (eight matrices, one launch)
        bool ok = tb.add(raw->mlp_2_weight, D, D, D, bw + L.wt_m2, D) && tb.add(raw->mlp_0_weight, E, D, E, bw + L.wt_m0, D) &&
                  tb.add(raw->clip_attn_out_proj_weight, E, E, E, bw + L.wt_out, E);
        for (int t = 0; t < 3; ++t)
            ok = ok && tb.add((const char*)raw->clip_attn_in_proj_weight + (size_t)t * E * E * 2, E, E, E, bw + L.wt_in + (size_t)t * E * E * 2, E);
        ok = ok && tb.add(raw->k_proj_1_2_weight, E, E, E, bw + L.wt_2, E) && tb.add(raw->v_proj_1_2_weight, E, E, E, bw + L.wt_2 + (size_t)E * E * 2, E);
        if (!ok) { set_error("tp_backward: transpose batch table too small"); return TP_ERR_LAUNCH; }
        TP_TRY(bw_transpose_batch_launch(MT, GT, tb, stream));
    }
    float* ln_g = (float*)(bw + L.ln_g);
    float* ln_b = (float*)(bw + L.ln_b);
    const void* gam_src[3] = {raw->ln_q_1_weight, raw->ln_k_1_weight, raw->ln_v_1_weight};
    const void* bet_src[3] = {raw->ln_q_1_bias, raw->ln_k_1_bias, raw->ln_v_1_bias};
    {
        BatchOps ops{};                                  // (six 1 K-element casts: one launch)
        for (int t = 0; t < 3; ++t) { ops.add(BATCH_OP_TO_F32, gam_src[t], ln_g + t * E, E); ops.add(BATCH_OP_TO_F32, bet_src[t], ln_b + t * E, E); }
        TP_TRY(pack_batch_launch(MT, ops, stream, nullptr));
    }

    // ---- the incoming gradient: as it is, or (fp16 chain) scaled by a dynamic power of two and cast ----------------------
    // (ONE pass over dy gives mlp[2]'s bias gradient — column sums of the unscaled dy — and max |dy| for the scale)
    const void* dyc = dy;                               // what the chain reads (dtype GT)
    {
        int n_amax = 0;
        const int slices = bw_colsum_rows_launch(MT, dy, D, Rq, D, colpart, stream, f16_chain ? (float*)(bw + L.amaxpart) : nullptr, &n_amax);
        if (slices < 0) return slices;
        TP_TRY(bw_reduce_many_parts_launch(MT, colpart, D, slices, D, grads->mlp_2_bias, redscratch, stream, nullptr));
        if (f16_chain) {
            TP_TRY(bw_scale_from_partials_launch((const float*)(bw + L.amaxpart), n_amax, scale, stream));
            TP_TRY(bw_scale_cast_launch(MT, dy, (long long)Rq * D, scale, bw + L.dy16, stream, sat));
            dyc = bw + L.dy16;
        }
    }
    // ---- mlp[2] ---------------------------------------------------------------------------------------------
    if (inplace) {
        TP_TRY(wgrad_rows(dyc, D, Rq, Rqp, D, D, fw + W.a2, D, TP_F16, bw + L.xt, grads->mlp_2_weight));
    } else {
        TP_TRY(T(GT, dyc, D, Rq, D, bw + L.dyT, Rqp));
        TP_TRY(T(TP_F16, fw + W.a2, D, Rq, D, bw + L.xt, Rqp));
        TP_TRY(wgrad(bw + L.dyT, bw + L.xt, D, D, Rqp, grads->mlp_2_weight));
    }
    TP_TRY(dgrad(dyc, D, Rq, D, bw + L.wt_m2, D, bw + L.dz2, D, TP_LINEAR_GELU_BWD, fw + W.z2, D));
    // ---- mlp[0] ---------------------------------------------------------------------------------------------
    if (inplace) {
        TP_TRY(bias_grad_rows(bw + L.dz2, D, Rq, D, grads->mlp_0_bias, GT, inv_scale));
        TP_TRY(wgrad_rows(bw + L.dz2, D, Rq, Rqp, D, E, fw + W.a1, E, TP_F16, bw + L.xt, grads->mlp_0_weight));
    } else {
        TP_TRY(T(GT, bw + L.dz2, D, Rq, D, bw + L.dz2T, Rqp, nullptr, nullptr, nullptr, colpart));
        TP_TRY(bias_grad(Rqp, D, grads->mlp_0_bias));
        TP_TRY(T(TP_F16, fw + W.a1, E, Rq, E, bw + L.xt, Rqp));
        TP_TRY(wgrad(bw + L.dz2T, bw + L.xt, D, E, Rqp, grads->mlp_0_weight));
    }
    TP_TRY(dgrad(bw + L.dz2, D, Rq, D, bw + L.wt_m0, E, bw + L.da1, E));
    // ---- out_proj ---------------------------------------------------------------------------------------------
    TP_TRY(bias_grad_rows(bw + L.da1, E, Rq, E, grads->clip_attn_out_proj_bias, GT, inv_scale));
    TP_TRY(wgrad_rows(bw + L.da1, E, Rq, Rqp, E, E, fw + W.o, E, TP_F16, bw + L.xt, grads->clip_attn_out_proj_weight));
    TP_TRY(dgrad(bw + L.da1, E, Rq, E, bw + L.wt_out, E, bw + L.dO, E));
    // ---- region attention ---------------------------------------------------------------------------------------
    TP_TRY(bw_region_attention_launch(GT, fw + W.q, fw + W.kv, fw + W.kv + kvE * 2, bw + L.dO, bw + L.dQ, bw + L.dKV,
                                      bw + L.dKV + kvE * 2, B, g, s, stream, sat));
    // ---- attention in-projection (rows of in_proj_weight / in_proj_bias: q | k | v) --------------------------------
    char* g_inw = (char*)grads->clip_attn_in_proj_weight;
    char* g_inb = (char*)grads->clip_attn_in_proj_bias;
    // Per LayerNorm'd input (q, then k, v): bias gradient of its in-projection rows, dgrad through the in-projection, the
    // LayerNorm backward — which also leaves the LayerNorm's OUTPUT row-major in the gradient dtype (the in-projection's weight
    // gradient reads it in place: no transposing / normalising pass) and the column sums of its dx (the bias gradient of the layer
    // in front of the LayerNorm) —, then the in-projection's weight gradient.
    float* lnpart = (float*)(bw + L.lnpart);
    const int nb = 256;
    char* const xn = bw + L.xt;                          // the LayerNorm output of the moment [rows, E] (GT); consumed by the next launch
    //   q
    TP_TRY(bias_grad_rows(bw + L.dQ, E, Rq, E, g_inb, GT, inv_scale));
    TP_TRY(dgrad(bw + L.dQ, E, Rq, E, bw + L.wt_in, E, bw + L.dq1, E));
    TP_TRY(bw_ln_backward_launch(GT, bw + L.dq1, fw + W.q1pre, (const float*)(fw + W.mr_q), ln_g, bw + L.dQ1pre, lnpart, nb, Rq, stream,
                                 ln_b, xn, sat));
    TP_TRY(bw_reduce_many_parts_launch(MT, lnpart, 3 * E, nb, E, grads->ln_q_1_weight, redscratch, stream, inv_scale));
    TP_TRY(bw_reduce_many_parts_launch(MT, lnpart + E, 3 * E, nb, E, grads->ln_q_1_bias, redscratch, stream, inv_scale));
    TP_TRY(wgrad_rows(bw + L.dQ, E, Rq, Rqp, E, E, xn, E, GT, nullptr, g_inw));
    //   k, v
    {
        void* gw[2] = {grads->ln_k_1_weight, grads->ln_v_1_weight};
        void* gb[2] = {grads->ln_k_1_bias, grads->ln_v_1_bias};
        void* gb2[2] = {grads->k_proj_1_2_bias, grads->v_proj_1_2_bias};
        for (int t = 0; t < 2; ++t) {
            const char* dX = bw + L.dKV + (size_t)t * kvE * 2;
            // (in_proj_bias, v third: colsum(dV) = colsum(dO) — dv_j = p_j dO and every head's p sums to 1 over a region's keys — a
            // pass over the 4 x smaller [Rq, E] instead of [R, E]; k third: colsum(dK), mathematically zero, summed as it is stored)
            if (t == 1) TP_TRY(bias_grad_rows(bw + L.dO, E, Rq, E, g_inb + (size_t)2 * E * 2, GT, inv_scale));
            else TP_TRY(bias_grad_rows(dX, E, R, E, g_inb + (size_t)(1 + t) * E * 2, GT, inv_scale));
            TP_TRY(dgrad(dX, E, R, E, bw + L.wt_in + (size_t)(1 + t) * E * E * 2, E, bw + L.dkv1 + (size_t)t * kvE * 2, E));
            TP_TRY(bw_ln_backward_launch(GT, bw + L.dkv1 + (size_t)t * kvE * 2, fw + W.h2 + (size_t)t * kvE * 2,
                                         (const float*)(fw + W.mr_kv) + (size_t)t * R * 2, ln_g + (1 + t) * E,
                                         bw + L.dH2 + (size_t)t * kvE * 2, lnpart, nb, R, stream, ln_b + (1 + t) * E, xn, sat));
            TP_TRY(bw_reduce_many_parts_launch(MT, lnpart, 3 * E, nb, E, gw[t], redscratch, stream, inv_scale));
            TP_TRY(bw_reduce_many_parts_launch(MT, lnpart + E, 3 * E, nb, E, gb[t], redscratch, stream, inv_scale));
            TP_TRY(bw_reduce_many_parts_launch(MT, lnpart + 2 * E, 3 * E, nb, E, gb2[t], redscratch, stream, inv_scale));   // colsum(dH2)
            TP_TRY(wgrad_rows(dX, E, R, Rp, E, E, xn, E, GT, nullptr, g_inw + (size_t)(1 + t) * E * E * 2));
        }
    }
    // ---- q_proj_1 (no bias) ---------------------------------------------------------------------------------------
    TP_TRY(wgrad_rows(bw + L.dQ1pre, E, Rq, Rqp, E, E, fw + W.q0, E, TP_F16, bw + L.xt, grads->q_proj_1_weight));
    // ---- k/v_proj_1[2] -----------------------------------------------------------------------------------------------
    {
        void* gw[2] = {grads->k_proj_1_2_weight, grads->v_proj_1_2_weight};     // (the bias gradients: colsum(dH2) from the LayerNorm backward above)
        for (int t = 0; t < 2; ++t) {
            const char* dH = bw + L.dH2 + (size_t)t * kvE * 2;
            char* hT = bw + L.xt;
            TP_TRY(wgrad_rows(dH, E, R, Rp, E, E, fw + W.hkv + (size_t)t * E * 2, 2 * E, TP_F16, hT, gw[t]));
            // (dZ1 leaves the chain in the MODEL dtype, still scaled by S: x_multi is in that dtype, and the first layer's weight
            // gradient reads both in place)
            TP_TRY(dgrad(dH, E, R, E, bw + L.wt_2 + (size_t)t * E * E * 2, E, bw + L.dZ1 + (size_t)t * E * 2, 2 * E,
                         TP_LINEAR_GELU_BWD, fw + W.z1 + (size_t)t * E * 2, 2 * E, MT));
        }
    }
    // ---- k/v_proj_1[0] -----------------------------------------------------------------------------------------------
    // dW0 [2E, 4096] = dZ1^T · x_multi (rows 0..E-1: k_proj_1[0], E..2E-1: v_proj_1[0]).  Both operands are in the model
    // dtype and row-major, so the contraction reads them in place (K-major GEMM operands): no dZ1^T, no x_multi^T.
    const WgradX XM{x_multi, xm_strides[1], N, xm_strides[0], xm_parts, kMulti / 4};
    const bool xm_tt = wgrad_tt_supported(R, 2 * E, kMulti, XM, 2 * E);
    if (!xm_tt && L.xmT == kNoSlab) {                   // (cannot happen for arguments tp_forward_train accepted)
        set_error("tp_backward: x_multi strides rule out the in-place weight gradient this workspace was sized for");
        return TP_ERR_INVALID_ARG;
    }
    if (xm_tt) {
        const int slices = bw_colsum_rows_launch(MT, bw + L.dZ1, 2 * E, R, 2 * E, colpart, stream);
        if (slices < 0) return slices;
        TP_TRY(bw_reduce_many_parts_launch(MT, colpart, 2 * E, slices, E, grads->k_proj_1_0_bias, redscratch, stream, inv_scale));
        TP_TRY(bw_reduce_many_parts_launch(MT, colpart + E, 2 * E, slices, E, grads->v_proj_1_0_bias, redscratch, stream, inv_scale));
        int* ctr = (counters && launch_no < 64) ? counters + 32 * launch_no++ : nullptr;
        return wgrad_tt_launch(MT, bw + L.dZ1, 2 * E, XM, R, 2 * E, kMulti, part, L.part_bytes, MT, grads->k_proj_1_0_weight,
                               ctr, stream, grads->v_proj_1_0_weight, E, inv_scale);
    }
    TP_TRY(bw_transpose_launch(MT, MT, bw + L.dZ1, 2 * E, R, 0, R, 2 * E, bw + L.dZ1T, Rp, Rp, nullptr, nullptr, nullptr, colpart, stream));
    TP_TRY(bw_reduce_many_parts_launch(MT, colpart, 2 * E, Rp / 64, E, grads->k_proj_1_0_bias, redscratch, stream, inv_scale));
    TP_TRY(bw_reduce_many_parts_launch(MT, colpart + E, 2 * E, Rp / 64, E, grads->v_proj_1_0_bias, redscratch, stream, inv_scale));
    if (xm_parts) {                                     // four [B, N, 1024] sources -> rows part*1024 .. of x_multi^T
        for (int i = 0; i < 4; ++i)
            TP_TRY(bw_transpose_launch(MT, MT, xm_parts[i], xm_strides[1], N, xm_strides[0], R, kMulti / 4,
                                       bw + L.xmT + (size_t)i * (kMulti / 4) * Rp * 2, Rp, Rp, nullptr, nullptr, nullptr,
                                       nullptr, stream));
    } else {
        TP_TRY(bw_transpose_launch(MT, MT, x_multi, xm_strides[1], N, xm_strides[0], R, kMulti, bw + L.xmT, Rp, Rp, nullptr,
                                   nullptr, nullptr, nullptr, stream));
    }
    {   // dW0 [2E, 4096] = dZ1^T · x_multi: rows 0..E-1 belong to k_proj_1[0], E..2E-1 to v_proj_1[0]
        const int Nout = 2 * E, Kin = kMulti;
        int S = 2;                                                       // 128 tiles of 256^2 -> 2 splits fill the chip
        GemmArgs a = plain_gemm(bw + L.dZ1T, Rp, bw + L.xmT, part, Kin, Nout, Kin, Rp / S, nullptr, 0);
        a.ldw_bytes = (long long)Rp * 2;
        a.groups = S; a.a_gs = (long long)(Rp / S) * 2; a.w_gs = (long long)(Rp / S) * 2; a.c_gs = (long long)Nout * Kin * 4;
        TP_TRY(launch(MT, TP_F32, a));
        TP_TRY(bw_reduce_parts_launch(MT, part, (long long)Nout * Kin, S, (long long)E * Kin, grads->k_proj_1_0_weight, stream, inv_scale));
        TP_TRY(bw_reduce_parts_launch(MT, part + (size_t)E * Kin, (long long)Nout * Kin, S, (long long)E * Kin,
                                      grads->v_proj_1_0_weight, stream, inv_scale));
    }
    return TP_OK;
}

int tp_backward(const tp_desc* desc, const void* x_multi, const int64_t xm_strides[3], const tp_weights* raw,
                const void* packed_weights, const void* train_workspace, const void* dy, const tp_grads* grads,
                void* bw_workspace, size_t bw_workspace_bytes, void* stream) {
    return backward_impl(desc, x_multi, nullptr, xm_strides, raw, packed_weights, train_workspace, dy, grads,
                         bw_workspace, bw_workspace_bytes, stream);
}

int tp_backward_parts(const tp_desc* desc, const void* const xm_parts[4], const int64_t part_strides[3],
                      const tp_weights* raw, const void* packed_weights, const void* train_workspace, const void* dy,
                      const tp_grads* grads, void* bw_workspace, size_t bw_workspace_bytes, void* stream) {
    if (!xm_parts) { set_error("tp_backward_parts: xm_parts is NULL"); return TP_ERR_INVALID_ARG; }
    return backward_impl(desc, nullptr, xm_parts, part_strides, raw, packed_weights, train_workspace, dy, grads,
                         bw_workspace, bw_workspace_bytes, stream);
}

}  // extern "C"
