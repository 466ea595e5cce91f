// Device-side pieces shared by the MFMA GEMM kernels of libtokenpacker_hip.so (gfx950 only):
// the MFMA wrapper, the erf-form GELU and the fused epilogue (LayerNorm-fold, bias, GELU, row
// statistics for the next LayerNorm, cast + packed stores).
#pragma once
#include "tp_internal.h"
#include <type_traits>

namespace tp {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// erf-form GELU (nn.GELU() default).  erf by Abramowitz–Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32
// round-off class next to the 1.0 it is added to): 1 rcp + 1 exp + ~12 FMA-class ops per element instead
// of ocml erff's ~45 with divergent branches — the epilogue runs with the matrix pipe idle, so its VALU
// time is pure cost (measured: ~0.1 ms per GELU layer at B=256 with erff).
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);      // exp(-x^2)
    const float erf_abs = fmaf(-poly, e, 1.0f);                                  // erf(|v|/sqrt2)
    const float half_v = 0.5f * v;
    return fmaf(half_v, copysignf(erf_abs, v), half_v);                          // 0.5 v (1 + erf)
}

constexpr int BK = 64;             // K-slab in elements
constexpr int ROW_BYTES = BK * 2;  // 128 B of K per tile row

// XCD-aware tile order (bijective): XCD x (= bid % 8 by dispatch order) owns a contiguous range of the
// tile list, so the tiles that share an A row-panel hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Fused epilogue of one BMxBN block tile held as 16x16 accumulator fragments:
//   acc[i][j][r] = C[m0 + wm*WM + i*16 + (lane&15)][n0 + wn*WN + j*16 + (lane>>4)*4 + r]
// (operands are swapped into the MFMA, so a lane owns 4 CONSECUTIVE columns of one row).
// mean_rstd[i] is the LayerNorm-fold (mean, rstd) of row i's fragment row (ignored without LN_FOLD).
// `smem` must be free (no DMA in flight, nobody reading) — the ROW_STATS path re-uses it after a barrier.
template <typename TO, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[WM / 16][WN / 16], const GemmArgs& p, const int g,
                                              const int m0, const int n0, const int tile_n, const int wm,
                                              const int wn, const int lane, const int tid,
                                              const float2 (&mean_rstd)[WM / 16], char* smem) {
    constexpr bool OUT_F32 = std::is_same<TO, float>::value;
    constexpr int NWM = BM / WM, NWN = BN / WN, NW = NWM * NWN;
    constexpr int FM = WM / 16, FN = WN / 16;
    const int flags = p.flags;
    const int col_base = n0 + wn * WN + (lane >> 4) * 4;
    const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
    const float* __restrict__ colsum = p.colsum ? p.colsum + g * p.colsum_gs : nullptr;
    char* __restrict__ Cg = p.C + g * p.c_gs;

    f32x4 bias_v[FN], csum_v[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        bias_v[j] = bias ? *(const f32x4*)(bias + col_base + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        csum_v[j] = (flags & TP_LINEAR_LN_FOLD) ? *(const f32x4*)(colsum + col_base + j * 16)
                                                : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    float rs1[FM], rs2[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 16 + (lane & 15);
        const bool row_ok = m < p.M;
        const float mu = mean_rstd[i].x, rstd = mean_rstd[i].y;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            f32x4 v = acc[i][j];
            if (flags & TP_LINEAR_LN_FOLD) v = rstd * (v - mu * csum_v[j]);
            v += bias_v[j];
            if (flags & TP_LINEAR_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            const long long coff = (long long)m * p.ldc + col_base + j * 16;
            if constexpr (OUT_F32) {
                if (row_ok) *(f32x4*)((float*)Cg + coff) = v;
            } else {
                using O4 = typename Vec<TO>::x4;
                if constexpr (std::is_same<TO, f16_t>::value) {     // saturate instead of producing inf
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], -65504.f), 65504.f);
                }
                const O4 o = __builtin_convertvector(v, O4);
                if (row_ok) *(O4*)((TO*)Cg + coff) = o;
                if (flags & TP_LINEAR_ROW_STATS) v = __builtin_convertvector(o, f32x4);  // stats of the ROUNDED values
            }
            if (flags & TP_LINEAR_ROW_STATS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1 += v[r]; s2 += v[r] * v[r]; }
            }
        }
        rs1[i] = s1; rs2[i] = s2;
    }

    if (flags & TP_LINEAR_ROW_STATS) {
        // reduce over the 4 lane groups that share a row, then over the NWN waves through LDS
        float* red = (float*)smem;                 // [NWN][BM][2]
        __syncthreads();                           // everyone is done with the K-slab buffers
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float s1 = rs1[i], s2 = rs2[i];
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (lane < 16) {
                const int rr = wm * WM + i * 16 + lane;
                red[(wn * BM + rr) * 2 + 0] = s1;
                red[(wn * BM + rr) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        // one slab per 128 output columns, whatever the tile: the partial-sum tree (lane -> 4 lane groups ->
        // the 128/WN waves of a slab) is identical for every tile shape, so results do not depend on the
        // tile the batch size selects (bit-exact batch invariance).
        constexpr int SLABS = BN / 128, WPS = 128 / WN;
        for (int idx = tid; idx < BM * SLABS; idx += NW * 64) {
            const int rr = idx % BM, sl = idx / BM;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WPS; ++w) {
                s1 += red[((sl * WPS + w) * BM + rr) * 2];
                s2 += red[((sl * WPS + w) * BM + rr) * 2 + 1];
            }
            const int m = m0 + rr;
            if (m < p.M) {
                float* so = p.stats_out + g * p.stats_out_gs + ((long long)(tile_n * SLABS + sl) * p.M + m) * 2;
                *(float2*)so = make_float2(s1, s2);
            }
        }
    }
}

}  // namespace tp
