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
// Householder QR in fp64 on the device, one launch pair per column (the matrices are 1024 x 1024, three of them — K side, V side,
// query side — batched through blockIdx.y): pack time only, ~tens of ms, never on the forward's path.
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

// Householder vector of column j (rows j .. E-1) of matrix blockIdx.x: v, beta = 2 / v^T v; the column itself becomes (alpha, 0, …).
__global__ void __launch_bounds__(1024)
qr_vec_kernel(double* __restrict__ Aall, double* __restrict__ Vall, double* __restrict__ betas, const int j) {
    __shared__ double red[16];
    double* A = Aall + (long long)blockIdx.x * QE * QP;
    double* V = Vall + (long long)blockIdx.x * QE;
    const int r = threadIdx.x;
    const double x = r >= j ? A[(long long)r * QP + j] : 0.0;
    const double norm2 = block_sum_1024(x * x, red);
    __shared__ double s_xj;
    if (r == j) s_xj = x;
    __syncthreads();
    const double xj = s_xj;
    const double alpha = xj >= 0.0 ? -sqrt(norm2) : sqrt(norm2);
    const double vj = xj - alpha;
    const double vtv = norm2 - xj * xj + vj * vj;
    if (r >= j) {
        V[r] = r == j ? vj : x;
        A[(long long)r * QP + j] = r == j ? alpha : 0.0;
    }
    if (r == 0) betas[blockIdx.x] = vtv > 0.0 ? 2.0 / vtv : 0.0;
}

// A[j.., c] -= beta v (v^T A[j.., c]) for the columns c > j (c = E: the bias column).  16 columns x 16 row slices per block: 128-byte
// row segments, and 64 blocks per matrix while the trailing block is wide (the factorisation is latency-bound on ~1000 dependent
// launch pairs; with 64-column blocks only 16 workgroups per matrix were in flight).
__global__ void __launch_bounds__(256)
qr_apply_kernel(double* __restrict__ Aall, const double* __restrict__ Vall, const double* __restrict__ betas, const int j) {
    __shared__ double part[16][17];
    double* A = Aall + (long long)blockIdx.y * QE * QP;
    const double* V = Vall + (long long)blockIdx.y * QE;
    const double beta = betas[blockIdx.y];
    const int cl = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int c = j + 1 + blockIdx.x * 16 + cl;
    const bool ok = c <= QE;
    double w = 0.0;
    if (ok)
        for (int r = j + rs; r < QE; r += 16) w = fma(V[r], A[(long long)r * QP + c], w);
    part[rs][cl] = w;
    __syncthreads();
    w = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) w += part[i][cl];           // fixed order: the factor does not depend on the launch geometry
    w *= beta;
    if (ok)
        for (int r = j + rs; r < QE; r += 16) A[(long long)r * QP + c] -= w * V[r];
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
                      f16_t* __restrict__ out, const float* __restrict__ d, float* __restrict__ d_out, int* __restrict__ sat) {
    const int n = blockIdx.x;
    const float cn = c[n];
    for (int k = threadIdx.x; k < QE; k += blockDim.x) {
        float pv = P[(long long)n * QE + k];
        if (P2) pv += P2[(long long)n * QE + k];
        float v = fmaf(-cn, wbar[k], pv);
        if (!(fabsf(v) <= 65504.f)) { if (sat) atomicAdd(sat, 1); v = fminf(fmaxf(v, -65504.f), 65504.f); }
        out[(long long)n * QE + k] = (f16_t)v;
    }
    if (threadIdx.x == 0 && d_out) d_out[n] = (d ? d[n] : 0.f) - cn * wbar[QE];
}

}  // namespace

size_t pack_qr_scratch_bytes(int nmat) { return (size_t)nmat * ((size_t)QE * QP * 8 + (size_t)QE * 8) + 256; }

// `A` of matrix m: scratch + m * E * (E + 1) doubles; behind the nmat matrices: the nmat Householder vectors, then the betas
static double* qr_mat(void* scratch, int m) { return (double*)scratch + (size_t)m * QE * QP; }

int pack_qr_center_launch(const void* w2_f16, const float* b2, void* scratch, int m, float* wbar, hipStream_t stream) {
    hipLaunchKernelGGL(qr_center_kernel, dim3(QP), dim3(1024), 0, stream, (const f16_t*)w2_f16, b2, qr_mat(scratch, m), wbar);
    return check_launch("qr_center_kernel");
}

int pack_qr_factor_launch(void* scratch, int nmat, hipStream_t stream) {
    double* A = qr_mat(scratch, 0);
    double* V = A + (size_t)nmat * QE * QP;
    double* betas = V + (size_t)nmat * QE;
    for (int j = 0; j < QE - 1; ++j) {
        hipLaunchKernelGGL(qr_vec_kernel, dim3(nmat), dim3(1024), 0, stream, A, V, betas, j);
        hipLaunchKernelGGL(qr_apply_kernel, dim3((QE - j + 15) / 16, nmat), dim3(256), 0, stream, A, V, betas, j);
    }
    return check_launch("qr_apply_kernel");
}

int pack_qr_extract_launch(const void* scratch, int m, void* r_f16, float* ctil, hipStream_t stream, int* sat) {
    hipLaunchKernelGGL(qr_extract_kernel, dim3(QE), dim3(256), 0, stream, (const double*)qr_mat((void*)scratch, m), (f16_t*)r_f16, ctil, sat);
    return check_launch("qr_extract_kernel");
}

int pack_center_product_launch(const float* P, const float* c, const float* wbar, void* out_f16, const float* d, float* d_out,
                               hipStream_t stream, int* sat, const float* P2) {
    hipLaunchKernelGGL(center_product_kernel, dim3(QE), dim3(256), 0, stream, P, P2, c, wbar, (f16_t*)out_f16, d, d_out, sat);
    return check_launch("center_product_kernel");
}

}  // namespace tp
