// tp_pack_qr.hip — pack-time factorisation behind the TRIANGULAR statistics GEMM (TP_TUNE_TRI_STATS, round 3).
//
// On the fused LayerNorm chain the layer in front of a LayerNorm, H2 = W2 h + b2, is computed for its row statistics ONLY
// (tp_api.hip).  The mean can be folded away altogether — centring is linear: with P = I - 11^T/E,
//     H2 - mean(H2) = P H2 = W2c h + b2c,   W2c = P W2 (every column of W2 minus its mean),  b2c = b2 - mean(b2),
// so the consumer's LayerNorm-folded GEMM uses Wc' = W'·W2c and d' = W'·b2c and needs no mean at all.  What is left to compute
// per row is  Σ_n (W2c h + b2c)_n^2 = || R h + c~ ||^2  with the QR factorisation  W2c = Q R,  c~ = Q^T b2c  (Q orthogonal, so
// the norm is unchanged): R is UPPER TRIANGULAR — output tile n0 of the statistics GEMM needs only the K-tiles k >= n0, 40 of
// the 64 (N-tile, K-tile) pairs at E = 1024 — and the sum is one of squares: no cancellation however large the row's mean was.
//
// Householder QR in fp64 on the device, blocked 16 columns at a time (one launch pair per panel; the matrices are 1024 x 1024, three
// of them — K side, V side, query side — batched through blockIdx.y): pack time only, never on the forward's path.
#include "tp_internal.h"

namespace tp {

namespace {

constexpr int QE = kEmbed;             // matrix order
static_assert(QE == 1024, "one thread per row in qr_center_kernel / qr_vec_kernel (1024-thread blocks)");
constexpr int QP = kEmbed + 1;         // row pitch: E columns of W2c + the b2c column carried through the reflections

__device__ __forceinline__ double block_sum_1024(double v, double* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    __syncthreads();
    return s;
}

// The factorisation's working matrix is COLUMN-major in the scratch: At[k][n], k <= E (k = E: the bias column) — a thread per row
// then reads a column with unit stride, and a wave can own whole columns (qr_apply_kernel).
// One block per column k: At[k][n] = w2[n][k] - mean_n w2[n][k] in fp64; wbar[k] = that mean (fp32).
__global__ void __launch_bounds__(1024)
qr_center_kernel(const f16_t* __restrict__ w2, const float* __restrict__ b2, double* __restrict__ A, float* __restrict__ wbar) {
    __shared__ double red[16];
    const int k = blockIdx.x, n = threadIdx.x;
    double x;
    if (k < QE) x = (double)(float)w2[(long long)n * QE + k];
    else x = b2 ? (double)b2[n] : 0.0;
    const double mean = block_sum_1024(x, red) / QE;
    A[(long long)k * QE + n] = x - mean;
    if (n == 0) wbar[k] = (float)mean;                  // wbar[E] = mean(b2)
}

// Blocked Householder QR (round 4): the reflections are generated and applied QB columns at a time, two launches per panel
// instead of two per COLUMN (the unblocked form was ~2000 dependent launches that each swept the 8 MB trailing matrix: 28 ms
// per pack, a third of all GPU time in a bench trace).  Same reflectors, same arithmetic per element — only grouped:
//   qr_panel_kernel   one workgroup per matrix, one thread per row, the row's QB panel values in registers: for each panel column
//                     the Householder vector (v, beta) — written to V[jj] with zeros above the diagonal — and its application to
//                     the panel's remaining columns.  All the column's dot products travel through ONE halving butterfly per
//                     wave (17 shuffles for 16 sums) and one cross-wave step.
//   qr_apply_kernel   a WAVE per two trailing columns (the bias column E included), the columns in registers, the panel's QB
//                     reflectors staged in LDS and applied one after the other: no barrier inside, the matrix read and
//                     written once per panel instead of once per reflection.
// (Two earlier forms of the same blocking, kept out: the panel in LDS with one dot product per wave was LDS-bandwidth-bound —
// 65 us per panel; sixteen independent shuffle reductions per wave per column were latency-bound — 178 us per panel.)
constexpr int QB = 16;                 // panel width

// ---- cross-lane sums without LDS traffic: DPP inside a row of 16 lanes, v_permlane{16,32}_swap (new on gfx950) across rows.
// (__shfl_xor is ds_bpermute: with 16 waves reducing 16 values each, the LDS pipe was what both kernels waited for.) ----
template <int CTRL>
__device__ __forceinline__ double qr_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHR4 = 0x114, DPP_ROW_ROR8 = 0x128;
__device__ __forceinline__ double qr_xor1(double v) { return qr_dpp<DPP_QUAD_XOR1>(v); }
__device__ __forceinline__ double qr_xor2(double v) { return qr_dpp<DPP_QUAD_XOR2>(v); }
__device__ __forceinline__ double qr_xor8(double v) { return qr_dpp<DPP_ROW_ROR8>(v); }
// lane l <- lane l ^ 4: a shift by four towards whichever side the partner sits on
__device__ __forceinline__ double qr_xor4(double v, bool bit2) {
    const double dn = qr_dpp<DPP_ROW_SHR4>(v), up = qr_dpp<DPP_ROW_SHL4>(v);    // lane l <- l - 4 | l + 4
    return bit2 ? dn : up;
}
// lanes with bit 5 clear: a(l) + a(l ^ 32); the others: b(l ^ 32) + b(l)   (one swap per dword, no select)
__device__ __forceinline__ double qr_halve32(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// lanes with bit 4 clear: a(l) + a(l ^ 16); the others: b(l ^ 16) + b(l)
__device__ __forceinline__ double qr_halve16(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// the sum over a wave's 64 lanes of each of 16 values: afterwards lane l holds the total of value qr_butterfly_slot(l)
__device__ __forceinline__ int qr_butterfly_slot(int lane) {
    return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}
__device__ __forceinline__ double qr_butterfly16(const double (&p)[16], int lane) {
    double q8[8], q4[4], q2[2];
    const bool h3 = lane & 8, h2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) q8[i] = qr_halve32(p[i], p[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q4[i] = qr_halve16(q8[i], q8[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) q2[i] = (h3 ? q4[i + 2] : q4[i]) + qr_xor8(h3 ? q4[i] : q4[i + 2]);
    double q = (h2 ? q2[1] : q2[0]) + qr_xor4(h2 ? q2[0] : q2[1], h2);
    q += qr_xor2(q);
    q += qr_xor1(q);
    return q;
}

// 256 threads, thread t owns rows t + 256 i (i < 4): four waves keep the cross-lane and LDS work per column small (with a thread
// per row, sixteen waves spent 2.7 us per column in the LDS pipe).
constexpr int QPT = 256, QPR = QE / QPT;

struct QrPanelShared {
    double red[(QPT / 64) * QB];                               // [wave][column]: the waves' partial dot products
    double dr[2][QB][2];                                       // [column]{x^T a_c, a_c[j]} (double-buffered: row j's entries are
};                                                             //  written for the next column while this one's are still read)

// column JJ of the panel (a template, not a loop: the compiler declines to unroll a body this size, and a rolled loop would index
// the register block dynamically — 528 B of scratch)
template <int JJ>
__device__ __forceinline__ void qr_panel_step(double (&a)[QPR][QB], QrPanelShared& sh, double* __restrict__ V,
                                              double* __restrict__ betas, const int j0) {
    const int t = threadIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = j0 + JJ;
    // (columns past E - 2 carry no reflection: v = 0, beta = 0 — the arithmetic below then leaves everything as it is)
    const bool live = j < QE - 1;
    double (*dr)[2] = sh.dr[JJ & 1];
    double x[QPR];
#pragma unroll
    for (int i = 0; i < QPR; ++i) x[i] = (live && t + QPT * i >= j) ? a[i][JJ] : 0.0;
    // With v = x - alpha e_j:  v^T a_c = x^T a_c - alpha a_c[j] — so the dot products need nothing of the Householder vector:
    // slot JJ = x^T x, slots c > JJ = x^T a_c, all in one butterfly; row j's own entries travel beside them.
    double p[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) {
        p[c] = 0.0;
        if (c >= JJ) {
#pragma unroll
            for (int i = 0; i < QPR; ++i) p[c] = fma(x[i], c == JJ ? x[i] : a[i][c], p[c]);
        }
    }
    const double q = qr_butterfly16(p, lane);
    if ((lane & 3) == 0) sh.red[wv * QB + qr_butterfly_slot(lane)] = q;
#pragma unroll
    for (int i = 0; i < QPR; ++i)
        if (t + QPT * i == j) {                                // row j's owner
#pragma unroll
            for (int c = JJ; c < QB; ++c) dr[c][1] = a[i][c];
        }
    __syncthreads();
    if (t < QB) {                                              // column t's partials, in a fixed order
        double s = sh.red[t];
#pragma unroll
        for (int w = 1; w < QPT / 64; ++w) s += sh.red[w * QB + t];
        dr[t][0] = s;
    }
    __syncthreads();
    const double norm2 = dr[JJ][0];
    const double xj = live ? dr[JJ][1] : 0.0;
    const double alpha = xj >= 0.0 ? -sqrt(norm2) : sqrt(norm2);
    const double vj = xj - alpha;
    const double vtv = norm2 - xj * xj + vj * vj;
    const double beta = (live && vtv > 0.0) ? 2.0 / vtv : 0.0;
    if (t == 0) betas[JJ] = beta;
    double bv[QPR];
#pragma unroll
    for (int i = 0; i < QPR; ++i) {
        const int r = t + QPT * i;
        const double v = !live ? 0.0 : (r == j ? vj : (r > j ? x[i] : 0.0));
        V[(long long)JJ * QE + r] = v;
        if (live) a[i][JJ] = r == j ? alpha : (r > j ? 0.0 : a[i][JJ]);
        bv[i] = beta * v;
    }
#pragma unroll
    for (int c = JJ + 1; c < QB; ++c) {
        const double wc = dr[c][0] - alpha * dr[c][1];
#pragma unroll
        for (int i = 0; i < QPR; ++i) a[i][c] -= bv[i] * wc;
    }
    if constexpr (JJ + 1 < QB) qr_panel_step<JJ + 1>(a, sh, V, betas, j0);
}

__global__ void __launch_bounds__(QPT)
qr_panel_kernel(double* __restrict__ Aall, double* __restrict__ Vall, double* __restrict__ betas, const int j0) {
    __shared__ QrPanelShared sh;
    double* A = Aall + (long long)blockIdx.x * QE * QP;
    double* V = Vall + (long long)blockIdx.x * QB * QE;
    const int t = threadIdx.x;
    double a[QPR][QB];
#pragma unroll
    for (int c = 0; c < QB; ++c)
#pragma unroll
        for (int i = 0; i < QPR; ++i) a[i][c] = (j0 + c < QE) ? A[(long long)(j0 + c) * QE + t + QPT * i] : 0.0;
    qr_panel_step<0>(a, sh, V, betas + blockIdx.x * QB, j0);
#pragma unroll
    for (int c = 0; c < QB; ++c)
#pragma unroll
        for (int i = 0; i < QPR; ++i)
            if (j0 + c < QE) A[(long long)(j0 + c) * QE + t + QPT * i] = a[i][c];
}

// The panel's QB reflectors applied, in order, to the trailing columns (the first: j0 + QB; c = E: the bias column).  A WAVE owns
// two columns — lane l holds rows l + 64 i of both in registers — so a reflector's dot products never leave the wave (no barrier,
// no partials in LDS, the sums by DPP / permlane swaps) and each LDS read of the reflector serves two columns.  Rows above the
// 64-row block of j0 are not touched (the panel's reflectors are zero there).  8 waves = 16 columns per workgroup — the fp64 FMAs
// of a reflector are what a CU spends its time on, so the columns are spread over as many CUs as there are; every sum's order is fixed.
constexpr int QAPPLY_WAVES = 8, QAPPLY_COLS = 2 * QAPPLY_WAVES;
__global__ void __launch_bounds__(64 * QAPPLY_WAVES)
qr_apply_kernel(double* __restrict__ Aall, const double* __restrict__ Vall, const double* __restrict__ betas, const int j0) {
    extern __shared__ double lds_qr[];                         // V panel [QB][QE]
    double* Vs = lds_qr;
    double* A = Aall + (long long)blockIdx.y * QE * QP;
    const double* V = Vall + (long long)blockIdx.y * QB * QE;
    const int i0 = j0 >> 6;                                    // first 64-row block a reflector of this panel reaches
    for (int i = i0 * 64 + threadIdx.x; i < QE; i += 64 * QAPPLY_WAVES) {
#pragma unroll
        for (int jj = 0; jj < QB; ++jj) Vs[jj * QE + i] = V[jj * QE + i];
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = j0 + QB + blockIdx.x * QAPPLY_COLS + 2 * wv, c1 = c0 + 1;
    const bool ok0 = c0 <= QE, ok1 = c1 <= QE;
    constexpr int NR = QE / 64;
    double a0[NR], a1[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        a0[i] = (ok0 && i >= i0) ? A[(long long)c0 * QE + lane + 64 * i] : 0.0;
        a1[i] = (ok1 && i >= i0) ? A[(long long)c1 * QE + lane + 64 * i] : 0.0;
    }
    __syncthreads();
    const bool h2 = lane & 4;
    for (int jj = 0; jj < QB; ++jj) {
        const double beta = betas[blockIdx.y * QB + jj];
        const double* v = Vs + jj * QE + lane;
        double vr[NR];
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) vr[i] = i >= i0 ? v[64 * i] : 0.0;
#pragma unroll
        for (int i = 0; i < NR; i += 2) {
            w0 = fma(vr[i], a0[i], w0);
            w1 = fma(vr[i], a1[i], w1);
            w2 = fma(vr[i + 1], a0[i + 1], w2);
            w3 = fma(vr[i + 1], a1[i + 1], w3);
        }
        // both sums over the wave: halve (lanes < 32 carry column 0, the others column 1), five plain steps, hand both to every lane
        double s = qr_halve32(w0 + w2, w1 + w3);
        s = qr_halve16(s, s);
        s += qr_xor8(s);
        s += qr_xor4(s, h2);
        s += qr_xor2(s);
        s += qr_xor1(s);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(s), (unsigned)__double2loint(s), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(s), (unsigned)__double2hiint(s), false, false);
        const double b0 = beta * __hiloint2double((int)hi[0], (int)lo[0]);    // lanes 0..31's total, in every lane
        const double b1 = beta * __hiloint2double((int)hi[1], (int)lo[1]);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            a0[i] -= b0 * vr[i];
            a1[i] -= b1 * vr[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        if (ok0 && i >= i0) A[(long long)c0 * QE + lane + 64 * i] = a0[i];
        if (ok1 && i >= i0) A[(long long)c1 * QE + lane + 64 * i] = a1[i];
    }
}

// R (upper triangle, fp16 — the statistics GEMM's weight, row-major, zeros below the diagonal) out of the column-major factor, through a
// 32 x 32 LDS tile; c~ = the transformed bias column (fp32).  grid (E/32, E/32), block (32, 8).
__global__ void __launch_bounds__(256)
qr_extract_kernel(const double* __restrict__ A, f16_t* __restrict__ r16, float* __restrict__ ctil, int* __restrict__ sat) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32, tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, n = n0 + tx;
        tile[ty + 8 * i][tx] = k >= n ? (float)A[(long long)k * QE + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        float v = tile[tx][ty + 8 * i];
        if (!(fabsf(v) <= 65504.f)) { if (sat) atomicAdd(sat, 1); v = fminf(fmaxf(v, -65504.f), 65504.f); }
        r16[(long long)n * QE + k] = (f16_t)v;
    }
    if (blockIdx.x == 0 && ty == 0 && ctil) ctil[n0 + tx] = (float)A[(long long)QE * QE + n0 + tx];
}

// P[n][k] (fp32 product W'·W2; P2: the product of the fold's residual, added when given) -> fp16( P[n][k] - c[n] wbar[k] ) = W'·W2c;
// d_out[n] = d[n] - c[n] bbar   (d may be NULL: no bias)
__global__ void __launch_bounds__(256)
center_product_kernel(const float* __restrict__ P, const float* __restrict__ P2, const float* __restrict__ c, const float* __restrict__ wbar,
                      f16_t* __restrict__ out, const float* __restrict__ d, float* __restrict__ d_out, int* __restrict__ sat,
                      f16_t* __restrict__ out3) {
    const int n = blockIdx.x;
    const float cn = c[n];
    for (int k = threadIdx.x; k < QE; k += blockDim.x) {
        float pv = P[(long long)n * QE + k];
        if (P2) pv += P2[(long long)n * QE + k];
        float v = fmaf(-cn, wbar[k], pv);
        if (!(fabsf(v) <= 65504.f)) { if (sat) atomicAdd(sat, 1); v = fminf(fmaxf(v, -65504.f), 65504.f); }
        const f16_t hi = (f16_t)v;
        out[(long long)n * QE + k] = hi;
        if (out3) {
            // rows of 2 E for the contraction over u with GemmArgs::a_k_dup = E: K-tile pairs (hi_t, lo_t) against u's K-tile t
            // (t = 0 .. 15) — [hi_0 lo_0 hi_1 lo_1 .. hi_15 lo_15]
            f16_t* o3 = out3 + (long long)n * 2 * QE;
            const int t = k >> 6, kk = k & 63;
            o3[(2 * t) * 64 + kk] = hi; o3[(2 * t + 1) * 64 + kk] = (f16_t)(v - (float)hi);
        }
    }
    if (threadIdx.x == 0 && d_out) d_out[n] = (d ? d[n] : 0.f) - cn * wbar[QE];
}

}  // namespace

size_t pack_qr_scratch_bytes(int nmat) { return (size_t)nmat * ((size_t)QE * QP * 8 + (size_t)QB * QE * 8 + QB * 8) + 256; }

// `At` of matrix m (column-major, E + 1 columns of E): scratch + m * E * (E + 1) doubles; behind the nmat matrices: the nmat panels of QB Householder vectors, then the betas
static double* qr_mat(void* scratch, int m) { return (double*)scratch + (size_t)m * QE * QP; }

int pack_qr_center_launch(const void* w2_f16, const float* b2, void* scratch, int m, float* wbar, hipStream_t stream) {
    hipLaunchKernelGGL(qr_center_kernel, dim3(QP), dim3(1024), 0, stream, (const f16_t*)w2_f16, b2, qr_mat(scratch, m), wbar);
    return check_launch("qr_center_kernel");
}

int pack_qr_factor_launch(void* scratch, int nmat, hipStream_t stream) {
    double* A = qr_mat(scratch, 0);
    double* V = A + (size_t)nmat * QE * QP;
    double* betas = V + (size_t)nmat * QB * QE;
    constexpr int lds = QB * QE * 8;
    static DynLdsAttr attr;                             // (per device, a failure is not cached: tp_internal.h)
    const hipError_t attr_err = attr.ensure(reinterpret_cast<const void*>(qr_apply_kernel), lds);
    if (attr_err != hipSuccess) { set_error("hipFuncSetAttribute(qr kernels, %d): %s", lds, hipGetErrorString(attr_err)); return TP_ERR_LAUNCH; }
    for (int j0 = 0; j0 < QE - 1; j0 += QB) {
        hipLaunchKernelGGL(qr_panel_kernel, dim3(nmat), dim3(QPT), 0, stream, A, V, betas, j0);
        const int trailing = QE + 1 - (j0 + QB);           // columns j0 + QB .. E (the bias column included)
        if (trailing > 0)
            hipLaunchKernelGGL(qr_apply_kernel, dim3((trailing + QAPPLY_COLS - 1) / QAPPLY_COLS, nmat), dim3(64 * QAPPLY_WAVES), lds, stream, A, V, betas, j0);
    }
    return check_launch("qr_apply_kernel");
}

int pack_qr_extract_launch(const void* scratch, int m, void* r_f16, float* ctil, hipStream_t stream, int* sat) {
    hipLaunchKernelGGL(qr_extract_kernel, dim3(QE / 32, QE / 32), dim3(32, 8), 0, stream, (const double*)qr_mat((void*)scratch, m), (f16_t*)r_f16, ctil, sat);
    return check_launch("qr_extract_kernel");
}

int pack_center_product_launch(const float* P, const float* c, const float* wbar, void* out_f16, const float* d, float* d_out,
                               hipStream_t stream, int* sat, const float* P2, void* out3_f16) {
    hipLaunchKernelGGL(center_product_kernel, dim3(QE), dim3(256), 0, stream, P, P2, c, wbar, (f16_t*)out_f16, d, d_out, sat, (f16_t*)out3_f16);
    return check_launch("center_product_kernel");
}

}  // namespace tp
