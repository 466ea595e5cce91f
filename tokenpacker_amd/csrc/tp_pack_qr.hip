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

// One block per column k (k = E: the bias column): A[n][k] = w2[n][k] - mean_n w2[n][k] in fp64; wbar[k] = that mean (fp32).
__global__ void __launch_bounds__(1024)
qr_center_kernel(const f16_t* __restrict__ w2, const float* __restrict__ b2, double* __restrict__ A, float* __restrict__ wbar) {
    __shared__ double red[16];
    const int k = blockIdx.x, n = threadIdx.x;
    double x;
    if (k < QE) x = (double)(float)w2[(long long)n * QE + k];
    else x = b2 ? (double)b2[n] : 0.0;
    const double mean = block_sum_1024(x, red) / QE;
    A[(long long)n * QP + k] = x - mean;
    if (n == 0) wbar[k] = (float)mean;                  // wbar[E] = mean(b2)
}

// Blocked Householder QR (round 4): the reflections are generated and applied QB columns at a time, two launches per panel
// instead of two per COLUMN (the unblocked form was ~2000 dependent launches that each swept the 8 MB trailing matrix: 28 ms
// per pack, a third of all GPU time in a bench trace).  Same reflectors, same arithmetic per element — only grouped:
//   qr_panel_kernel   one workgroup per matrix, one thread per row, the row's QB panel values in registers: for each panel column
//                     the Householder vector (v, beta) — written to V[jj] with zeros above the diagonal — and its application to
//                     the panel's remaining columns (all their dot products reduced in ONE block reduction per column);
//   qr_apply_kernel   one workgroup per 16 trailing columns (the bias column E included): its 1024 x 16 block in registers,
//                     the panel's QB reflectors staged in LDS and applied one after the other — the block is read and written
//                     once per panel instead of once per reflection.
constexpr int QB = 16;                 // panel width

__global__ void __launch_bounds__(1024)
qr_panel_kernel(double* __restrict__ Aall, double* __restrict__ Vall, double* __restrict__ betas, const int j0) {
    extern __shared__ double lds_qr[];                         // the panel, column-major [QB][QE] | red[16][QB] | broadcast slot
    double* Ap = lds_qr;
    double* red = lds_qr + QB * QE;
    double* s_bc = red + 16 * QB;
    double* A = Aall + (long long)blockIdx.x * QE * QP;
    double* V = Vall + (long long)blockIdx.x * QB * QE;
    const int r = threadIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = 0; c < QB; ++c) Ap[c * QE + r] = (j0 + c < QE) ? A[(long long)r * QP + j0 + c] : 0.0;
    // (a thread only ever touches its own row of the panel: no barrier needed around Ap)
    for (int jj = 0; jj < QB; ++jj) {
        const int j = j0 + jj;
        // (columns past E - 2 carry no reflection: v = 0, beta = 0 — the arithmetic below then leaves everything as it is)
        const bool live = j < QE - 1;
        const double ajj = Ap[jj * QE + r];
        const double x = (live && r >= j) ? ajj : 0.0;
        // ---- || x ||^2 and x_j ----
        double n2 = x * x;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n2 += __shfl_xor(n2, off);
        if (lane == 0) red[wv * QB] = n2;
        if (r == j) s_bc[0] = x;
        __syncthreads();
        double norm2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) norm2 += red[i * QB];
        const double xj = live ? s_bc[0] : 0.0;
        const double alpha = xj >= 0.0 ? -sqrt(norm2) : sqrt(norm2);
        const double vj = xj - alpha;
        const double vtv = norm2 - xj * xj + vj * vj;
        const double beta = (live && vtv > 0.0) ? 2.0 / vtv : 0.0;
        const double v = !live ? 0.0 : (r == j ? vj : (r > j ? x : 0.0));
        V[(long long)jj * QE + r] = v;
        if (r == 0) betas[blockIdx.x * QB + jj] = beta;
        if (live) Ap[jj * QE + r] = r == j ? alpha : (r > j ? 0.0 : ajj);
        __syncthreads();                                       // red / s_bc are re-used below
        // ---- the panel's remaining columns: w_c = beta v^T a_c, all of them behind ONE barrier ----
        for (int c = jj + 1; c < QB; ++c) {
            double pw = v * Ap[c * QE + r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) pw += __shfl_xor(pw, off);
            if (lane == 0) red[wv * QB + c] = pw;
        }
        __syncthreads();
        for (int c = jj + 1; c < QB; ++c) {
            double w = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) w += red[i * QB + c];   // fixed order
            Ap[c * QE + r] -= beta * w * v;
        }
        __syncthreads();
    }
    for (int c = 0; c < QB; ++c)
        if (j0 + c < QE) A[(long long)r * QP + j0 + c] = Ap[c * QE + r];
}

// The panel's QB reflectors applied, in order, to 16 trailing columns (first column j0 + QB + 16 blockIdx.x; c = E: the bias column).
// Thread (cl, rs): column cl of the block, rows rs, rs + 16, ...; the dot product of a reflector with a column is reduced over the
// 16 row slices through LDS in a fixed order.
__global__ void __launch_bounds__(256)
qr_apply_kernel(double* __restrict__ Aall, const double* __restrict__ Vall, const double* __restrict__ betas, const int j0) {
    extern __shared__ double lds_qr[];                         // V panel [QB][QE] | part[16][17]
    double* Vs = lds_qr;
    double (*part)[17] = (double (*)[17])(lds_qr + QB * QE);
    double* A = Aall + (long long)blockIdx.y * QE * QP;
    const double* V = Vall + (long long)blockIdx.y * QB * QE;
    for (int i = threadIdx.x; i < QB * QE; i += 256) Vs[i] = V[i];
    const int cl = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int c = j0 + QB + blockIdx.x * 16 + cl;
    const bool ok = c <= QE;
    // rows below j0 are not touched by this panel (its reflectors are zero there): start at the 16-aligned row below j0
    const int rbase = (j0 & ~15) + rs;
    constexpr int NR = QE / 16;
    double a[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = rbase + 16 * i;
        a[i] = (ok && r < QE) ? A[(long long)r * QP + c] : 0.0;
    }
    __syncthreads();
    for (int jj = 0; jj < QB; ++jj) {
        const double beta = betas[blockIdx.y * QB + jj];
        const double* v = Vs + jj * QE;
        double w = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rbase + 16 * i;
            if (r < QE) w = fma(v[r], a[i], w);
        }
        part[rs][cl] = w;
        __syncthreads();
        w = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) w += part[i][cl];         // fixed order: the factor does not depend on the launch geometry
        w *= beta;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rbase + 16 * i;
            if (r < QE) a[i] -= w * v[r];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = rbase + 16 * i;
        if (ok && r < QE) A[(long long)r * QP + c] = a[i];
    }
}

// R (upper triangle, fp16 — the statistics GEMM's weight, zeros below the diagonal) and c~ = the transformed bias column (fp32)
__global__ void __launch_bounds__(256)
qr_extract_kernel(const double* __restrict__ A, f16_t* __restrict__ r16, float* __restrict__ ctil, int* __restrict__ sat) {
    const int n = blockIdx.x;
    for (int k = threadIdx.x; k < QE; k += blockDim.x) {
        float v = k >= n ? (float)A[(long long)n * QP + k] : 0.f;
        if (!(fabsf(v) <= 65504.f)) { if (sat) atomicAdd(sat, 1); v = fminf(fmaxf(v, -65504.f), 65504.f); }
        r16[(long long)n * QE + k] = (f16_t)v;
    }
    if (threadIdx.x == 0 && ctil) ctil[n] = (float)A[(long long)n * QP + QE];
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
        if (out3) {                                            // [hi | hi | lo] rows of 3 E: the contraction over (u_hi | u_lo | u_hi)
            f16_t* o3 = out3 + (long long)n * 3 * QE + k;
            o3[0] = hi; o3[QE] = hi; o3[2 * QE] = (f16_t)(v - (float)hi);
        }
    }
    if (threadIdx.x == 0 && d_out) d_out[n] = (d ? d[n] : 0.f) - cn * wbar[QE];
}

}  // namespace

size_t pack_qr_scratch_bytes(int nmat) { return (size_t)nmat * ((size_t)QE * QP * 8 + (size_t)QB * QE * 8 + QB * 8) + 256; }

// `A` of matrix m: scratch + m * E * (E + 1) doubles; behind the nmat matrices: the nmat panels of QB Householder vectors, then the betas
static double* qr_mat(void* scratch, int m) { return (double*)scratch + (size_t)m * QE * QP; }

int pack_qr_center_launch(const void* w2_f16, const float* b2, void* scratch, int m, float* wbar, hipStream_t stream) {
    hipLaunchKernelGGL(qr_center_kernel, dim3(QP), dim3(1024), 0, stream, (const f16_t*)w2_f16, b2, qr_mat(scratch, m), wbar);
    return check_launch("qr_center_kernel");
}

int pack_qr_factor_launch(void* scratch, int nmat, hipStream_t stream) {
    double* A = qr_mat(scratch, 0);
    double* V = A + (size_t)nmat * QE * QP;
    double* betas = V + (size_t)nmat * QB * QE;
    constexpr int lds = (QB * QE + 16 * 17) * 8, lds_panel = (QB * QE + 16 * QB + 2) * 8;
    static hipError_t attr_err = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qr_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(qr_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_panel);
        return e;
    }();
    if (attr_err != hipSuccess) { set_error("hipFuncSetAttribute(qr kernels, %d): %s", lds, hipGetErrorString(attr_err)); return TP_ERR_LAUNCH; }
    for (int j0 = 0; j0 < QE - 1; j0 += QB) {
        hipLaunchKernelGGL(qr_panel_kernel, dim3(nmat), dim3(1024), lds_panel, stream, A, V, betas, j0);
        const int trailing = QE + 1 - (j0 + QB);           // columns j0 + QB .. E (the bias column included)
        if (trailing > 0)
            hipLaunchKernelGGL(qr_apply_kernel, dim3((trailing + 15) / 16, nmat), dim3(256), lds, stream, A, V, betas, j0);
    }
    return check_launch("qr_apply_kernel");
}

int pack_qr_extract_launch(const void* scratch, int m, void* r_f16, float* ctil, hipStream_t stream, int* sat) {
    hipLaunchKernelGGL(qr_extract_kernel, dim3(QE), dim3(256), 0, stream, (const double*)qr_mat((void*)scratch, m), (f16_t*)r_f16, ctil, sat);
    return check_launch("qr_extract_kernel");
}

int pack_center_product_launch(const float* P, const float* c, const float* wbar, void* out_f16, const float* d, float* d_out,
                               hipStream_t stream, int* sat, const float* P2, void* out3_f16) {
    hipLaunchKernelGGL(center_product_kernel, dim3(QE), dim3(256), 0, stream, P, P2, c, wbar, (f16_t*)out_f16, d, d_out, sat, (f16_t*)out3_f16);
    return check_launch("center_product_kernel");
}

}  // namespace tp
