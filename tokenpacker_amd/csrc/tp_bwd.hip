// tp_bwd.hip — the non-GEMM kernels of the projector's BACKWARD pass on gfx950 (stage-1
// training updates only the projector, reference llava/train/train.py:950-953).  The contractions of the
// backward pass run on the forward's MFMA kernels (tp_gemm8.hip / tp_gemm.hip):
//   dgrad  dX[M,K] = dY[M,N] · W[N,K]        =  linear(A = dY, W-operand = W^T [K,N])
//   wgrad  dW[N,K] = dY^T[N,M] · X[M,K]      =  linear(A = dY^T [N,M], W-operand = X^T [K,M]), split over M
// so what is needed here are transposed (and cast / LayerNorm-applied) operand copies, column sums for the
// bias gradients, the split-K reduction, the LayerNorm backward and the region-attention backward.
// Gradients travel in the MODEL dtype (bf16 range for bf16 training, fp16 with the caller's loss scaling),
// accumulate in fp32.  No atomics anywhere: every reduction is two-stage and ordered -> deterministic.
#include "tp_internal.h"

namespace tp {

template <typename T> __device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ T sat_cast(float v);
template <> __device__ __forceinline__ f16_t sat_cast<f16_t>(float v) { return (f16_t)fminf(fmaxf(v, -65504.f), 65504.f); }
template <> __device__ __forceinline__ bf16_t sat_cast<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ float sat_cast<float>(float v) { return v; }
// Sticky saturation report of the backward's fp16 chain (tp_backward's status word, bit `bit`): a wave in which any lane is about
// to clamp (|v| >= 65520: what IEEE rounding would have turned into an fp16 inf; NaN counts) ORs the bit — one ballot per call site
// and, when it fires, one atomic per wave.  `flag` NULL: nothing.
template <typename T> __device__ __forceinline__ void bw_sat_report(int* flag, int bit, float max_abs) {
    if constexpr (std::is_same<T, f16_t>::value) {
        if (flag && __builtin_amdgcn_ballot_w64(!(max_abs < 65520.f)) != 0 && (threadIdx.x & 63) == 0)
            __hip_atomic_fetch_or(flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------
// dst[c][r] = cast( f(src[r][c]) )  for r < R (zero for R <= r < Rpad), c < C.   f = identity, or the LayerNorm
// (src - mean_r) * rstd_r * gamma_c + beta_c when `mean_rstd` is given.  src rows may be batch-strided (the CLIP
// tower's [:,1:] slices).  Optionally emits per-row-block column sums of the SOURCE values (bias gradients):
// colsum_part[blockIdx.y][c].  64x64 tiles through LDS, 256 threads.
template <typename TS, typename TD>
__global__ void __launch_bounds__(256)
transpose_kernel(const TS* __restrict__ src, long long ld, int rows_per_batch, long long batch_stride, int R, int C,
                 TD* __restrict__ dst, long long ldd, int Rpad, const float* __restrict__ mean_rstd,
                 const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ colsum_part) {
    // 64 x 64 tile; every thread moves 16 elements with two 16-byte accesses on either side (rows are 16-byte
    // aligned and C, ld, ldd multiples of 8 — checked on the host).  The LDS tile is fp32 with a 65-word pitch:
    // the column gathers of the write phase hit 16 different banks.
    __shared__ float tile[64][65];
    using S8 = typename Vec<TS>::x8;
    using D8 = typename Vec<TD>::x8;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tr = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;       // 64 rows x 4 segments of 16
    {
        const int r = r0 + tr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + seg + h * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (r < R && c < C) {
                const int b = r / rows_per_batch;
                const S8 t = *(const S8*)(src + (long long)b * batch_stride + (long long)(r - b * rows_per_batch) * ld + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[tr][seg + h * 8 + e] = v[e];
        }
    }
    __syncthreads();
    if (colsum_part && threadIdx.x < 64) {                          // column sums of the raw source tile, row order
        float sum = 0.f;
        for (int rr = 0; rr < 64; ++rr) sum += tile[rr][threadIdx.x];
        if (c0 + (int)threadIdx.x < C) colsum_part[(long long)blockIdx.y * C + c0 + threadIdx.x] = sum;
    }
    {
        const int c = c0 + tr;                                       // output row = source column
        if (c < C) {
            const float gm = mean_rstd ? gamma[c] : 1.f, bt = mean_rstd ? beta[c] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rb = r0 + seg + h * 8;
                if (rb < Rpad) {
                    D8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = rb + e;
                        float v = tile[seg + h * 8 + e][tr];
                        if (mean_rstd && r < R) v = (v - mean_rstd[2 * (long long)r]) * mean_rstd[2 * (long long)r + 1] * gm + bt;
                        o[e] = sat_cast<TD>(r < R ? v : 0.f);
                    }
                    *(D8*)(dst + (long long)c * ldd + rb) = o;
                }
            }
        }
    }
}

template <typename TS, typename TD>
static int transpose_launch_t(const void* src, long long ld, int rpb, long long bstride, int R, int C, void* dst,
                              long long ldd, int Rpad, const float* mr, const float* gamma, const float* beta,
                              float* colsum_part, hipStream_t stream) {
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((Rpad + 63) / 64));
    hipLaunchKernelGGL((transpose_kernel<TS, TD>), grid, dim3(256), 0, stream, (const TS*)src, ld, rpb, bstride, R, C,
                       (TD*)dst, ldd, Rpad, mr, gamma, beta, colsum_part);
    return check_launch("transpose_kernel");
}

// The same 64 x 64 tiling for a BATCH of plain matrices in one launch (round 6: the eight weights the dgrad GEMMs read transposed —
// 7 us each at a 32-image step, launch latency mostly): blockIdx.z picks the matrix, blocks beyond its tile grid leave at once.
template <typename TS, typename TD>
__global__ void __launch_bounds__(256)
transpose_batch_kernel(const TransposeBatch tb) {
    const TransposeOp o = tb.op[blockIdx.z];
    if ((int)blockIdx.x * 64 >= o.C || (int)blockIdx.y * 64 >= o.Rpad) return;
    __shared__ float tile[64][65];
    using S8 = typename Vec<TS>::x8;
    using D8 = typename Vec<TD>::x8;
    const TS* __restrict__ src = (const TS*)o.src;
    TD* __restrict__ dst = (TD*)o.dst;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tr = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
    {
        const int r = r0 + tr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + seg + h * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (r < o.R && c < o.C) {
                const S8 t = *(const S8*)(src + (long long)r * o.ld + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[tr][seg + h * 8 + e] = v[e];
        }
    }
    __syncthreads();
    const int c = c0 + tr;
    if (c < o.C) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rb = r0 + seg + h * 8;
            if (rb < o.Rpad) {
                D8 out;
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = sat_cast<TD>(rb + e < o.R ? tile[seg + h * 8 + e][tr] : 0.f);
                *(D8*)(dst + (long long)c * o.ldd + rb) = out;
            }
        }
    }
}

int bw_transpose_batch_launch(int src_dtype, int dst_dtype, const TransposeBatch& tb, hipStream_t stream) {
    if (tb.count <= 0) return TP_OK;
    if (tb.count > kTransposeBatch) { set_error("bw transpose batch: %d matrices > %d", tb.count, kTransposeBatch); return TP_ERR_INVALID_ARG; }
    int gx = 1, gy = 1;
    for (int i = 0; i < tb.count; ++i) {
        const TransposeOp& o = tb.op[i];
        if ((o.C & 7) || (o.ld & 7) || (o.ldd & 7) || (o.Rpad & 7) || ((uintptr_t)o.src & 15) || ((uintptr_t)o.dst & 15)) {
            set_error("bw transpose batch: matrix %d is not 8-element / 16-byte aligned", i);
            return TP_ERR_INVALID_ARG;
        }
        gx = std::max(gx, (o.C + 63) / 64); gy = std::max(gy, (o.Rpad + 63) / 64);
    }
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)tb.count);
#define TP_TRB(SD, DD, TS, TD) if (src_dtype == SD && dst_dtype == DD) { hipLaunchKernelGGL((transpose_batch_kernel<TS, TD>), grid, dim3(256), 0, stream, tb); return check_launch("transpose_batch_kernel"); }
    TP_TRB(TP_BF16, TP_BF16, bf16_t, bf16_t) TP_TRB(TP_F16, TP_F16, f16_t, f16_t) TP_TRB(TP_BF16, TP_F16, bf16_t, f16_t) TP_TRB(TP_F16, TP_BF16, f16_t, bf16_t)
#undef TP_TRB
    set_error("bw transpose batch: unsupported dtypes %d -> %d", src_dtype, dst_dtype);
    return TP_ERR_INVALID_ARG;
}

int bw_transpose_launch(int src_dtype, int dst_dtype, const void* src, long long ld, int rows_per_batch,
                        long long batch_stride, int R, int C, void* dst, long long ldd, int Rpad, const float* mean_rstd,
                        const float* gamma, const float* beta, float* colsum_part, hipStream_t stream) {
    if (rows_per_batch <= 0 || rows_per_batch > R) { rows_per_batch = R > 0 ? R : 1; batch_stride = 0; }
#define TP_TR(SD, DD, TS, TD) if (src_dtype == SD && dst_dtype == DD) return transpose_launch_t<TS, TD>(src, ld, \
        rows_per_batch, batch_stride, R, C, dst, ldd, Rpad, mean_rstd, gamma, beta, colsum_part, stream)
    TP_TR(TP_BF16, TP_BF16, bf16_t, bf16_t); TP_TR(TP_F16, TP_F16, f16_t, f16_t);
    TP_TR(TP_F16, TP_BF16, f16_t, bf16_t);   TP_TR(TP_BF16, TP_F16, bf16_t, f16_t);
#undef TP_TR
    set_error("bw transpose: unsupported dtypes %d -> %d", src_dtype, dst_dtype);
    return TP_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------------------------------
// out[i] = cast( scale-free sum over parts of part[s][i] ), i < n   (split-K wgrad partials, column-sum partials)
template <typename TD>
__global__ void __launch_bounds__(256)
reduce_parts_kernel(const float* __restrict__ part, long long part_stride, int nparts, long long n, TD* __restrict__ out,
                    const float* __restrict__ out_scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long long)k * part_stride + i];
    if (out_scale) s *= out_scale[0];                   // (a power of two: exact)
    out[i] = sat_cast<TD>(s);
}

// Same sum, many parts (column-sum / LayerNorm partials: hundreds to thousands of parts of a few thousand values),
// in two stages so that the whole chip streams the partials with 16-byte loads: stage 1 — grid (n / 256 column
// blocks, S part slices), a thread owns 4 consecutive columns and every 4th part of its slice, the 4 part-lanes
// combine through LDS in a fixed order -> scratch[S][n]; stage 2 — reduce_parts_kernel over the S slices.  The
// summation tree depends on (nparts, S) only: deterministic.
__global__ void __launch_bounds__(256)
reduce_cols_stage1_kernel(const float* __restrict__ part, long long part_stride, int nparts, int n,
                          float* __restrict__ scratch) {
    __shared__ f32x4 red[4][64];
    const int cq = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cq) * 4;
    const int S = gridDim.y, slice = blockIdx.y;
    const int per = (nparts + S - 1) / S;
    const int k0 = slice * per, k1 = (k0 + per < nparts) ? k0 + per : nparts;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < n) {
        const float* b = part + c;
        int k = k0 + pl;
        for (; k + 12 < k1; k += 16) {                   // four independent 16-byte loads in flight per thread
            s0 += *(const f32x4*)(b + (long long)k * part_stride);
            s1 += *(const f32x4*)(b + (long long)(k + 4) * part_stride);
            s2 += *(const f32x4*)(b + (long long)(k + 8) * part_stride);
            s3 += *(const f32x4*)(b + (long long)(k + 12) * part_stride);
        }
        for (; k < k1; k += 4) s0 += *(const f32x4*)(b + (long long)k * part_stride);
    }
    red[pl][cq] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (pl == 0 && c < n)
        *(f32x4*)(scratch + (long long)slice * n + c) = (red[0][cq] + red[1][cq]) + (red[2][cq] + red[3][cq]);
}

int bw_reduce_many_parts_launch(int dst_dtype, const float* part, long long part_stride, int nparts, int n, void* out,
                                float* scratch, hipStream_t stream, const float* out_scale) {
    if ((n & 3) || (part_stride & 3) || ((uintptr_t)part & 15)) {
        set_error("bw reduce: n / part stride must be multiples of 4 floats, partials 16-byte aligned");
        return TP_ERR_INVALID_ARG;
    }
    int S = kReduceSlices;
    while (S > 1 && nparts < 8 * S) S >>= 1;              // at least 8 parts per slice
    hipLaunchKernelGGL(reduce_cols_stage1_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)S), dim3(256), 0, stream,
                       part, part_stride, nparts, n, scratch);
    if (int rc = check_launch("reduce_cols_stage1_kernel")) return rc;
    return bw_reduce_parts_launch(dst_dtype, scratch, n, S, n, out, stream, out_scale);
}

// Column sums of a row-major 16-bit matrix (bias gradients of the layers whose weight gradient reads dY in place):
// part[slice][c] = sum over the slice's rows of src[r][c].  A thread owns 8 consecutive columns (one 16-byte load per
// row) and every 4th row of its slice; the 4 row-lanes combine through LDS in a fixed order.
template <typename TS>
__global__ void __launch_bounds__(256)
colsum_rows_kernel(const TS* __restrict__ src, long long ld, long long R, int C, float* __restrict__ part,
                   float* __restrict__ amax_part) {
    using S8 = typename Vec<TS>::x8;
    __shared__ float red[4][64][8];
    __shared__ float red_max[256];
    float amax = 0.f;                                     // (amax_part: max |v| of the block's values, NaN sticks — bw_dynamic_scale)
    const int co = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + co) * 8;
    const int S = gridDim.y, slice = blockIdx.y;
    const long long per = (R + S - 1) / S;
    const long long r0 = slice * per, r1 = (r0 + per < R) ? r0 + per : R;
    float acc[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[u][e] = 0.f;
    if (c < C) {
        const TS* b = src + c;
        long long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {                   // four independent 16-byte loads in flight per thread
            S8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const S8*)(b + (r + 4 * u) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[u][e];
                    acc[u][e] += f;
                    if (amax_part) { const float a = fabsf(f); amax = (a != a || a > amax) ? a : amax; }
                }
        }
        for (; r < r1; r += 4) {
            const S8 v = *(const S8*)(b + r * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                acc[0][e] += f;
                if (amax_part) { const float a = fabsf(f); amax = (a != a || a > amax) ? a : amax; }
            }
        }
    }
    if (amax_part) {
        red_max[threadIdx.x] = amax;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { const float a = red_max[threadIdx.x + o], m = red_max[threadIdx.x]; red_max[threadIdx.x] = (a != a || a > m) ? a : m; }
            __syncthreads();
        }
        if (threadIdx.x == 0) amax_part[blockIdx.y * gridDim.x + blockIdx.x] = red_max[0];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][co][e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            part[(long long)slice * C + c + e] = (red[0][co][e] + red[1][co][e]) + (red[2][co][e] + red[3][co][e]);
    }
}

// -> part[slices][C]; returns the number of slices written (a fixed function of R and C), or a negative status
int bw_colsum_rows_launch(int dtype, const void* src, long long ld, long long R, int C, float* part, hipStream_t stream,
                          float* amax_part, int* amax_count) {
    if ((C & 7) || (ld & 7) || ((uintptr_t)src & 15)) {
        set_error("bw colsum: columns / row stride must be multiples of 8 elements, the source 16-byte aligned");
        return TP_ERR_INVALID_ARG;
    }
    const int gx = (C + 511) / 512;
    long long S = 512 / gx;                              // ~512 workgroups
    if (S > R / 64) S = R / 64;
    if (S < 1) S = 1;
    if (S > kColsumMaxSlices) S = kColsumMaxSlices;
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(colsum_rows_kernel<bf16_t>, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, stream, (const bf16_t*)src, ld, R, C, part, amax_part);
    else if (dtype == TP_F16)
        hipLaunchKernelGGL(colsum_rows_kernel<f16_t>, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, stream, (const f16_t*)src, ld, R, C, part, amax_part);
    else { set_error("bw colsum: unsupported dtype %d", dtype); return TP_ERR_INVALID_ARG; }
    if (int rc = check_launch("colsum_rows_kernel")) return rc;
    if (amax_count) *amax_count = (int)(gx * S);          // (amax_part entries written: kColsumAmaxParts bounds it)
    return (int)S;
}

int bw_reduce_parts_launch(int dst_dtype, const float* part, long long part_stride, int nparts, long long n, void* out,
                           hipStream_t stream, const float* out_scale) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dst_dtype == TP_BF16)
        hipLaunchKernelGGL(reduce_parts_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, part, part_stride, nparts, n, (bf16_t*)out, out_scale);
    else if (dst_dtype == TP_F16)
        hipLaunchKernelGGL(reduce_parts_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, part, part_stride, nparts, n, (f16_t*)out, out_scale);
    else if (dst_dtype == TP_F32)
        hipLaunchKernelGGL(reduce_parts_kernel<float>, dim3(blocks), dim3(256), 0, stream, part, part_stride, nparts, n, (float*)out, out_scale);
    else { set_error("bw reduce: unsupported dtype %d", dst_dtype); return TP_ERR_INVALID_ARG; }
    return check_launch("reduce_parts_kernel");
}

// ---------------------------------------------------------------------------------------------------
// Dynamic power-of-two scale of a gradient tensor (tp_internal.h: bw_scale_from_partials_launch).  The per-workgroup max |v| come
// from the column-sum pass that reads dy anyway (colsum_rows_kernel's amax_part; a NaN sticks through the comparisons); one
// workgroup reduces them and writes S = 2^k with amax * S in [16, 32) and 1 / S (S = 1 for an all-zero or non-finite tensor).
__global__ void __launch_bounds__(256)
scale_from_amax_kernel(const float* __restrict__ part, int nparts, float* __restrict__ scale) {
    __shared__ float red[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) { const float a = part[i]; m = (a != a || a > m) ? a : m; }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { const float a = red[threadIdx.x + o], b = red[threadIdx.x]; red[threadIdx.x] = (a != a || a > b) ? a : b; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float amax = red[0];
        float S = 1.f;
        if (amax > 0.f && amax < INFINITY) {            // (zero, inf, NaN: S = 1 — the values travel as they are)
            int e;
            (void)frexpf(amax, &e);                      // amax = f * 2^e, f in [0.5, 1)  ->  amax * 2^(5 - e) in [16, 32)
            int k = 5 - e;
            k = k > 60 ? 60 : (k < -60 ? -60 : k);
            S = ldexpf(1.f, k);
        }
        scale[0] = S; scale[1] = 1.f / S;
    }
}
template <typename TS>
__global__ void __launch_bounds__(256)
scale_cast_f16_kernel(const TS* __restrict__ src, long long n8, const float* __restrict__ scale, f16_t* __restrict__ dst,
                      int* __restrict__ sat_flag) {
    using S8 = typename Vec<TS>::x8;
    using D8 = typename Vec<f16_t>::x8;
    const float S = scale[0];
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const S8 v = *(const S8*)(src + i * 8);
        D8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e] * S, a = fabsf(f); mx = (a != a || a > mx) ? a : mx; o[e] = sat_cast<f16_t>(f); }
        *(D8*)(dst + i * 8) = o;
    }
    bw_sat_report<f16_t>(sat_flag, 1, mx);              // (an inf / NaN in dy, which no finite scale brings into range)
}

int bw_scale_from_partials_launch(const float* amax_part, int nparts, float* scale, hipStream_t stream) {
    hipLaunchKernelGGL(scale_from_amax_kernel, dim3(1), dim3(256), 0, stream, amax_part, nparts, scale);
    return check_launch("scale_from_amax_kernel");
}

int bw_scale_cast_launch(int src_dtype, const void* src, long long n, const float* scale, void* dst_f16, hipStream_t stream, int* sat_flag) {
    if ((n & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst_f16 & 15)) { set_error("bw scale cast: element count must be a multiple of 8, pointers 16-byte aligned"); return TP_ERR_INVALID_ARG; }
    const int nb = 2048;
    if (src_dtype == TP_BF16)
        hipLaunchKernelGGL(scale_cast_f16_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)src, n / 8, scale, (f16_t*)dst_f16, sat_flag);
    else if (src_dtype == TP_F16)
        hipLaunchKernelGGL(scale_cast_f16_kernel<f16_t>, dim3(nb), dim3(256), 0, stream, (const f16_t*)src, n / 8, scale, (f16_t*)dst_f16, sat_flag);
    else { set_error("bw scale cast: unsupported dtype %d", src_dtype); return TP_ERR_INVALID_ARG; }
    return check_launch("scale_cast_f16_kernel");
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward over rows of E = 1024 (one wave per row, 16 elements per lane):
//   xhat = (x - mean) rstd,  g = dy gamma,  dx = rstd (g - mean(g) - xhat mean(g xhat))
// plus per-workgroup partial sums of dgamma = sum_rows dy xhat, dbeta = sum_rows dy and (round 6) the column sums of dx —
// the bias gradient of the linear layer in front of the LayerNorm, which used to be a pass of its own over dx —
// (part[blk][0][E], part[blk][1][E], part[blk][2][E]; reduced by bw_reduce_many_parts).  x: fp16 (forward activations), dy/dx: TG.
// `xn` (round 6, optional): the LayerNorm's OUTPUT xhat gamma + beta, row-major in TG — the activation operand of the weight
// gradient of the layer BEHIND the LayerNorm, which the K-major GEMM then reads in place (it used to be produced by a
// transposing pass that re-read x and fetched (mean, rstd) per element: 0.82 ms per B = 256 step for the three LayerNorms).
template <typename TG>
__global__ void __launch_bounds__(256)
ln_backward_kernel(const TG* __restrict__ dy, const f16_t* __restrict__ x, const float* __restrict__ mean_rstd,
                   const float* __restrict__ gamma, TG* __restrict__ dx, float* __restrict__ part, long long rows,
                   const float* __restrict__ beta, TG* __restrict__ xn, int* __restrict__ sat_flag) {
    constexpr int E = kEmbed;
    __shared__ float red[3][4][E];                       // 48 KiB
    float sat_mx = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[16], db[16], gm[16], dxs[16], bt[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        dg[e] = 0.f; db[e] = 0.f; dxs[e] = 0.f; gm[e] = gamma[(e >> 3) * 512 + lane * 8 + (e & 7)];
        bt[e] = xn ? beta[(e >> 3) * 512 + lane * 8 + (e & 7)] : 0.f;
    }
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        const float mu = mean_rstd[2 * r], rstd = mean_rstd[2 * r + 1];
        float dyv[16], xh[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            using G8 = typename Vec<TG>::x8;
            const G8 d8 = *(const G8*)(dy + r * E + h * 512 + lane * 8);
            const f16x8 x8 = *(const f16x8*)(x + r * E + h * 512 + lane * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { dyv[h * 8 + e] = (float)d8[e]; xh[h * 8 + e] = ((float)x8[e] - mu) * rstd; }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float g = dyv[e] * gm[e]; s1 += g; s2 = fmaf(g, xh[e], s2); }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
        const float m1 = s1 * (1.0f / E), m2 = s2 * (1.0f / E);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            using G8 = typename Vec<TG>::x8;
            G8 o, n8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = h * 8 + e;
                const float dxv = rstd * (dyv[k] * gm[k] - m1 - xh[k] * m2);
                { const float a = fabsf(dxv); sat_mx = (a != a || a > sat_mx) ? a : sat_mx; }
                o[e] = sat_cast<TG>(dxv);
                dxs[k] += (float)o[e];                  // (the ROUNDED value: what a column-sum pass over dx would have read)
                dg[k] = fmaf(dyv[k], xh[k], dg[k]);
                db[k] += dyv[k];
                n8[e] = sat_cast<TG>(fmaf(xh[k], gm[k], bt[k]));
            }
            *(G8*)(dx + r * E + h * 512 + lane * 8) = o;
            if (xn) *(G8*)(xn + r * E + h * 512 + lane * 8) = n8;
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        red[0][wave][(e >> 3) * 512 + lane * 8 + (e & 7)] = dg[e];
        red[1][wave][(e >> 3) * 512 + lane * 8 + (e & 7)] = db[e];
        red[2][wave][(e >> 3) * 512 + lane * 8 + (e & 7)] = dxs[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * E; i += 256) {
        const int w = i / E, c = i - w * E;
        part[((long long)blockIdx.x * 3 + w) * E + c] = red[w][0][c] + red[w][1][c] + red[w][2][c] + red[w][3][c];
    }
    bw_sat_report<TG>(sat_flag, 2, sat_mx);
}

int bw_ln_backward_launch(int gdtype, const void* dy, const void* x_f16, const float* mean_rstd, const float* gamma,
                          void* dx, float* part, int nblocks, long long rows, hipStream_t stream, const float* beta, void* xn,
                          int* sat_flag) {
    if (xn && !beta) { set_error("bw ln backward: the normalised output needs beta"); return TP_ERR_INVALID_ARG; }
    if (gdtype == TP_BF16)
        hipLaunchKernelGGL(ln_backward_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, stream, (const bf16_t*)dy,
                           (const f16_t*)x_f16, mean_rstd, gamma, (bf16_t*)dx, part, rows, beta, (bf16_t*)xn, sat_flag);
    else
        hipLaunchKernelGGL(ln_backward_kernel<f16_t>, dim3(nblocks), dim3(256), 0, stream, (const f16_t*)dy,
                           (const f16_t*)x_f16, mean_rstd, gamma, (f16_t*)dx, part, rows, beta, (f16_t*)xn, sat_flag);
    return check_launch("ln_backward_kernel");
}

// ---------------------------------------------------------------------------------------------------
// Region-to-point attention backward.  One wavefront per coarse query, lane layout of the forward kernel
// (lane l: elements [8l, 8l+8) of head l>>4 and [512+8l, ...) of head 4 + (l>>4)).  With p = softmax_j(q·k_j scale):
//   dv_j = p_j dO,   dP_j = dO·v_j,   D = sum_j p_j dP_j,   dS_j = p_j (dP_j - D),
//   dq = scale sum_j dS_j k_j,        dk_j = scale dS_j q.
// Every key/value row belongs to exactly one query (regions partition the grid), so dK / dV rows are written
// once, without atomics.  Three sweeps over the region's keys (max+denominator | D and dV | dq and dK) keep the
// kernel valid for any s; K/V are fp16 forward activations, dO / dQ / dK / dV the gradient dtype TG.
template <typename TG>
__global__ void __launch_bounds__(256)
region_attention_bwd_kernel(const f16_t* __restrict__ q, const f16_t* __restrict__ k, const f16_t* __restrict__ v,
                            const TG* __restrict__ dout, TG* __restrict__ dq, TG* __restrict__ dk, TG* __restrict__ dv,
                            int B, int g, int s, float scale, int* __restrict__ sat_flag) {
    float sat_mx = 0.f;
    auto trk = [&](float v) __attribute__((always_inline)) { const float a = fabsf(v); sat_mx = (a != a || a > sat_mx) ? a : sat_mx; return v; };
    constexpr int E = kEmbed;
    using G8 = typename Vec<TG>::x8;
    const int lane = threadIdx.x & 63;
    const int G = g / s, M = G * G, N = g * g, S2 = s * s;
    const long long qi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= (long long)B * M) return;
    const int b = (int)(qi / M), m = (int)(qi % M);
    const int i = m / G, j = m % G;
    const int ea = lane * 8, eb = 512 + lane * 8;
    auto row_of = [&](int kk) -> long long {
        const int a = kk / s, c = kk - a * s;
        return ((long long)b * N + (i * s + a) * g + j * s + c) * E;
    };
    auto load8h = [&](const f16_t* p, float (&f)[8]) {
        const f16x8 t = *(const f16x8*)p;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)t[e];
    };
    auto dot16 = [&](float d) {                          // 128-wide head dot product: 16-lane butterfly
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) d += __shfl_xor(d, off);
        return d;
    };
    float qa[8], qb[8], doa[8], dob[8];
    load8h(q + qi * E + ea, qa);
    load8h(q + qi * E + eb, qb);
    {
        const G8 ta = *(const G8*)(dout + qi * E + ea), tb = *(const G8*)(dout + qi * E + eb);
#pragma unroll
        for (int e = 0; e < 8; ++e) { doa[e] = (float)ta[e]; dob[e] = (float)tb[e]; }
    }
    // sweep 1: max and denominator
    float mxa = -INFINITY, mxb = -INFINITY;
    for (int kk = 0; kk < S2; ++kk) {
        float ka[8], kb[8];
        const long long r = row_of(kk);
        load8h(k + r + ea, ka); load8h(k + r + eb, kb);
        float da = 0.f, db = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { da = fmaf(qa[e], ka[e], da); db = fmaf(qb[e], kb[e], db); }
        mxa = fmaxf(mxa, dot16(da) * scale); mxb = fmaxf(mxb, dot16(db) * scale);
    }
    float dena = 0.f, denb = 0.f;
    for (int kk = 0; kk < S2; ++kk) {
        float ka[8], kb[8];
        const long long r = row_of(kk);
        load8h(k + r + ea, ka); load8h(k + r + eb, kb);
        float da = 0.f, db = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { da = fmaf(qa[e], ka[e], da); db = fmaf(qb[e], kb[e], db); }
        dena += __expf(dot16(da) * scale - mxa); denb += __expf(dot16(db) * scale - mxb);
    }
    const float inva = 1.0f / dena, invb = 1.0f / denb;
    // sweep 2: D = sum_j p_j dP_j, and dV
    float Da = 0.f, Db = 0.f;
    for (int kk = 0; kk < S2; ++kk) {
        float ka[8], kb[8], va[8], vb[8];
        const long long r = row_of(kk);
        load8h(k + r + ea, ka); load8h(k + r + eb, kb);
        load8h(v + r + ea, va); load8h(v + r + eb, vb);
        float da = 0.f, db = 0.f, pa_ = 0.f, pb_ = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            da = fmaf(qa[e], ka[e], da); db = fmaf(qb[e], kb[e], db);
            pa_ = fmaf(doa[e], va[e], pa_); pb_ = fmaf(dob[e], vb[e], pb_);
        }
        const float pa = __expf(dot16(da) * scale - mxa) * inva, pb = __expf(dot16(db) * scale - mxb) * invb;
        Da = fmaf(pa, dot16(pa_), Da); Db = fmaf(pb, dot16(pb_), Db);
        G8 oa, ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) { oa[e] = sat_cast<TG>(pa * doa[e]); ob[e] = sat_cast<TG>(pb * dob[e]); }   // (|p| <= 1: cannot exceed dO)
        *(G8*)(dv + r + ea) = oa; *(G8*)(dv + r + eb) = ob;
    }
    // sweep 3: dS, dq, dK
    float dqa[8], dqb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dqa[e] = 0.f; dqb[e] = 0.f; }
    for (int kk = 0; kk < S2; ++kk) {
        float ka[8], kb[8], va[8], vb[8];
        const long long r = row_of(kk);
        load8h(k + r + ea, ka); load8h(k + r + eb, kb);
        load8h(v + r + ea, va); load8h(v + r + eb, vb);
        float da = 0.f, db = 0.f, pa_ = 0.f, pb_ = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            da = fmaf(qa[e], ka[e], da); db = fmaf(qb[e], kb[e], db);
            pa_ = fmaf(doa[e], va[e], pa_); pb_ = fmaf(dob[e], vb[e], pb_);
        }
        const float pa = __expf(dot16(da) * scale - mxa) * inva, pb = __expf(dot16(db) * scale - mxb) * invb;
        const float dsa = pa * (dot16(pa_) - Da) * scale, dsb = pb * (dot16(pb_) - Db) * scale;
        G8 oa, ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dqa[e] = fmaf(dsa, ka[e], dqa[e]); dqb[e] = fmaf(dsb, kb[e], dqb[e]);
            oa[e] = sat_cast<TG>(trk(dsa * qa[e])); ob[e] = sat_cast<TG>(trk(dsb * qb[e]));
        }
        *(G8*)(dk + r + ea) = oa; *(G8*)(dk + r + eb) = ob;
    }
    G8 oa, ob;
#pragma unroll
    for (int e = 0; e < 8; ++e) { oa[e] = sat_cast<TG>(trk(dqa[e])); ob[e] = sat_cast<TG>(trk(dqb[e])); }
    *(G8*)(dq + qi * E + ea) = oa; *(G8*)(dq + qi * E + eb) = ob;
    bw_sat_report<TG>(sat_flag, 4, sat_mx);
}

int bw_region_attention_launch(int gdtype, const void* q, const void* k, const void* v, const void* dout, void* dq,
                               void* dk, void* dv, int B, int grid, int s, hipStream_t stream, int* sat_flag) {
    const int G = grid / s, M = G * G;
    const long long nq = (long long)B * M;
    const unsigned blocks = (unsigned)((nq + 3) / 4);
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    if (gdtype == TP_BF16)
        hipLaunchKernelGGL(region_attention_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const f16_t*)q,
                           (const f16_t*)k, (const f16_t*)v, (const bf16_t*)dout, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, B, grid, s, scale, sat_flag);
    else
        hipLaunchKernelGGL(region_attention_bwd_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, (const f16_t*)q,
                           (const f16_t*)k, (const f16_t*)v, (const f16_t*)dout, (f16_t*)dq, (f16_t*)dk, (f16_t*)dv, B, grid, s, scale, sat_flag);
    return check_launch("region_attention_bwd_kernel");
}

}  // namespace tp
