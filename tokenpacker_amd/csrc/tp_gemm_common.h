// Device-side pieces shared by the MFMA GEMM kernels of libtokenpacker_hip.so (gfx950 only):
// the MFMA wrapper, the erf-form GELU and the fused epilogue (LayerNorm-fold, bias, GELU, row
// statistics for the next LayerNorm, cast + packed stores).
#pragma once
#include "tp_internal.h"
#include <type_traits>

#ifndef TP_EPI_PROBE
#define TP_EPI_PROBE 0         // timing probes of the epilogue (variant builds only; garbage results): 1 no output stores | 2 no arithmetic in front of them
#endif
#ifndef TP_EPI_FULL_LINE
#define TP_EPI_FULL_LINE 0     // A/B build flag (round 6): half-precision outputs as whole 128-byte lines per row (1; 2 = non-temporal)
#endif
#ifndef TP_EPI_LANE_ADJ
#define TP_EPI_LANE_ADJ 0      // A/B build flag (round 6): the half-precision output stores with adjacent lanes adjacent in memory
#endif

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

// erf-form GELU (nn.GELU() default) by Abramowitz–Stegun rational forms (|abs err| <= 1.5e-7 / 3e-7, i.e. fp32
// round-off class next to the 1.0 it is added to) instead of ocml erff's ~45 ops with divergent branches — the
// epilogue runs with the matrix pipe idle, so its VALU time is pure cost (measured: ~0.1 ms per GELU layer at
// B=256 with erff; 5.5 us per 256x256 tile with 7.1.26, profiles/r01w_ktile_fit.txt).
// d/dv of the erf-form GELU:  Phi(v) + v phi(v)   (7.1.26: the exp(-v^2/2) it needs is shared with the pdf term)
__device__ __forceinline__ float gelu_erf_grad(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);      // exp(-v^2/2)
    const float erf_abs = fmaf(-poly, e, 1.0f);
    const float cdf = fmaf(0.5f, copysignf(erf_abs, v), 0.5f);
    return fmaf(v * 0.3989422804014327f, e, cdf);                               // + v * exp(-v^2/2)/sqrt(2 pi)
}

__device__ __forceinline__ float gelu_erf(float v) {
    // Abramowitz & Stegun 7.1.28:  erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16,  |err| <= 3e-7 — one transcendental
    // (v_rcp_f32) per element instead of the rcp + exp of 7.1.26, and the Horner chain and the four squarings pack
    // two elements per v_pk_* instruction.  Overflow of the 16th power gives rcp(inf) = 0, i.e. erf = 1.
    const float x = fabsf(v) * 0.70710678118654752440f;
    float p = fmaf(0.0000430638f, x, 0.0002765672f);
    p = fmaf(p, x, 0.0001520143f);
    p = fmaf(p, x, 0.0092705272f);
    p = fmaf(p, x, 0.0422820123f);
    p = fmaf(p, x, 0.0705230784f);
    p = fmaf(p, x, 1.0f);
    p *= p; p *= p; p *= p; p *= p;
    const float r = __builtin_amdgcn_rcpf(p);                                    // 1 - erf(|v|/sqrt2)
    return fmaf(-0.5f * fabsf(v), r, fmaxf(v, 0.0f));                            // 0.5 v (1 + erf(v/sqrt2))
}

// the same on two elements at once, written on 2-vectors so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32 (two
// elements per instruction at the scalar instruction's issue cost) for the Horner chain and the squarings
typedef float f32x2_ev __attribute__((ext_vector_type(2)));
#ifndef TP_GELU_FAST
#define TP_GELU_FAST 0          // A/B build flag (round 6): 22 instead of 25 issue slots per pair (below)
#endif
#if TP_GELU_FAST
// The same approximation with three issue slots less per pair (22 for 25): |v| and a clamp at 9 in ONE v_min_f32 (source modifier; beyond 9 the
// correction term is < 1e-16 and the clamp keeps t^16 of BOTH lanes inside fp32 for the shared reciprocal), 1/sqrt2 folded into the coefficients,
// ONE v_rcp_f32 per pair (1/a = b * rcp(a b), 1/b = a * rcp(a b): the reciprocal is a quarter-rate instruction), relu as v_med3_f32.
__device__ __forceinline__ f32x2_ev gelu_erf2(f32x2_ev v) {
    // (inline asm: fminf / fmaxf / fmed3f on a value the compiler cannot prove canonical cost a canonicalising v_max_f32 each)
    f32x2_ev a;
    { float a0, a1;
      asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(a0) : "v"(v[0]), "v"(9.0f));
      asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(a1) : "v"(v[1]), "v"(9.0f));
      a[0] = a0; a[1] = a1; }
    constexpr float s = 0.70710678118654752440f, s2 = s * s, s3 = s2 * s, s4 = s2 * s2, s5 = s4 * s, s6 = s3 * s3;
    f32x2_ev p = __builtin_elementwise_fma(a, (f32x2_ev)(0.0000430638f * s6), (f32x2_ev)(0.0002765672f * s5));
    p = __builtin_elementwise_fma(p, a, (f32x2_ev)(0.0001520143f * s4));
    p = __builtin_elementwise_fma(p, a, (f32x2_ev)(0.0092705272f * s3));
    p = __builtin_elementwise_fma(p, a, (f32x2_ev)(0.0422820123f * s2));
    p = __builtin_elementwise_fma(p, a, (f32x2_ev)(0.0705230784f * s));
    p = __builtin_elementwise_fma(p, a, (f32x2_ev)(1.0f));
    p *= p; p *= p; p *= p; p *= p;
    const float R = __builtin_amdgcn_rcpf(p[0] * p[1]);
    const f32x2_ev r = f32x2_ev{p[1], p[0]} * R;
    f32x2_ev relu;
    { float r0, r1;
      asm("v_max_f32_e32 %0, 0, %1" : "=v"(r0) : "v"(v[0]));
      asm("v_max_f32_e32 %0, 0, %1" : "=v"(r1) : "v"(v[1]));
      relu[0] = r0; relu[1] = r1; }
    return __builtin_elementwise_fma(a * -0.5f, r, relu);
}
#else
__device__ __forceinline__ f32x2_ev gelu_erf2(f32x2_ev v) {
    const f32x2_ev av = __builtin_elementwise_abs(v);
    const f32x2_ev x = av * 0.70710678118654752440f;
    f32x2_ev p = __builtin_elementwise_fma(x, (f32x2_ev)(0.0000430638f), (f32x2_ev)(0.0002765672f));
    p = __builtin_elementwise_fma(p, x, (f32x2_ev)(0.0001520143f));
    p = __builtin_elementwise_fma(p, x, (f32x2_ev)(0.0092705272f));
    p = __builtin_elementwise_fma(p, x, (f32x2_ev)(0.0422820123f));
    p = __builtin_elementwise_fma(p, x, (f32x2_ev)(0.0705230784f));
    p = __builtin_elementwise_fma(p, x, (f32x2_ev)(1.0f));
    p *= p; p *= p; p *= p; p *= p;
    f32x2_ev r;
    r[0] = __builtin_amdgcn_rcpf(p[0]); r[1] = __builtin_amdgcn_rcpf(p[1]);
    // 0.5 v (1 + erf(v/sqrt2)) = relu(v) - 0.5 |v| (1 - erf(|v|/sqrt2)):  no cancellation in the negative tail
    // (relu as 0.5 (v + |v|): exact, and packed — fmaxf would cost a canonicalising v_max_f32 besides the max)
    return __builtin_elementwise_fma(av * -0.5f, r, (v + av) * 0.5f);
}
#endif

// GemmArgs::a_region_s: region-major row index -> raster row index (reference divide_feature's grouping, builder.py:96-105)
__device__ __forceinline__ int region_major_to_raster(int row, int g, int s) {
    const int N = g * g, S2 = s * s, G = g / s;
    const int b = row / N, t = row - b * N;
    const int q = t / S2, kk = t - q * S2;
    const int a = kk / s, c = kk - a * s;
    const int qi = q / G, qj = q - qi * G;
    return b * N + (qi * s + a) * g + qj * s + c;
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
// `smem`: 8 KiB of LDS scratch nobody else touches while the epilogue runs (ROW_STATS reduction).
// `lds_par` (optional): the tile's epilogue parameters staged in LDS by the caller — fp32 bias[BN] at +0 and
// colsum[BN] at +4*BN, indexed by the column INSIDE the tile — instead of global loads (persistent kernel: no
// ordinary vector load may be outstanding next to the LDS-DMA queue, or hipcc drains it).
// Barriers are raw s_barrier + lgkmcnt(0): a __syncthreads() would also drain vmcnt, i.e. the DMA of the NEXT tile.
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Sticky fp16-saturation report (GemmArgs::sat_flag): sat_track folds two values about to be clamped into the lane's running
// max |v| — one v_max3_f32 with |.| source modifiers per pair; sat_report ORs GemmArgs::sat_bit into the flag when any lane
// of the wave saw a value the reference's fp16 arithmetic would have turned into inf.
__device__ __forceinline__ float sat_track(float m, float a, float b) {
    return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));       // -> v_max3_f32 m, |a|, |b|
}
__device__ __forceinline__ void sat_report(const GemmArgs& p, float sat_max) {
    // 65520 = the smallest magnitude IEEE round-to-nearest turns into an fp16 inf (what the reference would have produced)
    if (p.sat_flag && __builtin_amdgcn_ballot_w64(!(sat_max < 65520.f)) != 0) {
        if ((threadIdx.x & 63) == 0)                   // (every lane is active here; mbcnt would be hoisted into a register the K loops lack)
            __hip_atomic_fetch_or(p.sat_flag, p.sat_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// TRAIN_EPI compiles in the two training-only epilogue features (TP_LINEAR_SAVE_PRE, TP_LINEAR_GELU_BWD); the
// inference kernels are instantiated without them so that their register budget is not taxed.
// STATS_ONLY (TP_LINEAR_NO_STORE): the row statistics of the unrounded result are the only output — no conversion, no store.
// LAZY_PAR (with LDS_PARAMS; tp_gemm_pair.hip): bias / colsum and the rows' (mean, rstd at lds_par + 8 BN + 8 row) are read from
// LDS at the point of use instead of up front — 48 registers less across the epilogue (same values, same arithmetic).
template <typename TO, int BM, int BN, int WM, int WN, bool LDS_PARAMS = false, bool TRAIN_EPI = false, bool STATS_ONLY = false,
          bool LAZY_PAR = false>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[WM / 16][WN / 16], const GemmArgs& p, const int g,
                                              const int m0, const int n0, const int tile_n, const int wm,
                                              const int wn, const int lane, const int tid,
                                              const float2 (&mean_rstd)[WM / 16], char* smem,
                                              const char* lds_par = nullptr) {
    constexpr bool OUT_F32 = std::is_same<TO, float>::value;
    constexpr int NWM = BM / WM, NWN = BN / WN, NW = NWM * NWN;
    constexpr int FM = WM / 16, FN = WN / 16;
    const int flags = p.flags;
    const int col_base = n0 + wn * WN + (lane >> 4) * 4;
    const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
    const float* __restrict__ colsum = p.colsum ? p.colsum + g * p.colsum_gs : nullptr;
    char* __restrict__ Cg = p.C + g * p.c_gs;
    int n0s = n0;                                      // the tile's first column inside its output slab (GemmArgs::c_split_cols)
    if (p.c_split_cols > 0 && n0 >= p.c_split_cols) { n0s = n0 - p.c_split_cols; Cg += p.c_split_stride_bytes; }
    // The output goes out through a buffer descriptor that ends after row M - 1: rows past the end are dropped by
    // the hardware range check, so every wave issues the SAME number of store instructions for every tile (the
    // persistent kernel's counted s_waitcnt at the tile seam relies on that) and no branch guards the stores.
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Cg, 0, (int)(unsigned)((long long)p.M * p.ldc * (long long)sizeof(TO)), 0x00020000);

    static_assert(!LAZY_PAR || LDS_PARAMS, "LAZY_PAR reads the parameters from LDS");
    f32x4 bias_v[FN], csum_v[FN];
    const int lc_lazy = wn * WN + (lane >> 4) * 4;     // column inside the tile
    if constexpr (LAZY_PAR) {
    } else if constexpr (LDS_PARAMS) {
        const int lc = wn * WN + (lane >> 4) * 4;      // column inside the tile
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bias_v[j] = *(const f32x4*)(lds_par + (lc + j * 16) * 4);
            csum_v[j] = *(const f32x4*)(lds_par + BN * 4 + (lc + j * 16) * 4);
        }
    } else {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bias_v[j] = bias ? *(const f32x4*)(bias + col_base + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
            csum_v[j] = (flags & TP_LINEAR_LN_FOLD) ? *(const f32x4*)(colsum + col_base + j * 16)
                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // Row statistics are accumulated per 64-column slice (VW slices per wave) and combined below in a fixed
    // tree, so a wave that owns 128 columns produces bit-identical results to two waves that own 64 each.
    // What travels is (mean, M2 = sum of squared deviations from that mean) of a group of values, merged pairwise by
    // Chan's formula — NOT (sum, sum of squares): var = E[x^2] - mean^2 in fp32 loses every digit once |mean| >> std
    // (rows with a large common offset, e.g. behind massive-activation channels of a trained CLIP), whereas the
    // merged M2 stays accurate to fp32 round-off of the spread itself.  A lane's own 4·FNV values are accumulated
    // as shifted sums around its first value (the shift makes the one-pass form safe).
    constexpr int VW = WN / 64;                    // 64-column slices per wave
    constexpr int FNV = FN / VW;                   // fragments per slice (= 4)
    static_assert(WN % 64 == 0, "wave tile must be a multiple of 64 columns");
    float rs1[FM][VW], rs2[FM][VW];
    float sat_max = 0.f;                          // largest |value| this lane handed to the fp16 clamp (GemmArgs::sat_flag)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 16 + (lane & 15);
        const bool row_ok = m < p.M;
        float mu = mean_rstd[i].x, rstd = mean_rstd[i].y;
        if constexpr (LAZY_PAR) {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t mr = *(const f32x2_t*)(lds_par + 2 * BN * 4 + (wm * WM + i * 16 + (lane & 15)) * 8);
            mu = mr[0]; rstd = mr[1];
        }
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) {
            float s1 = 0.f, s2 = 0.f, pivot = 0.f;
            // fragments are finished in PAIRS (j, j+1): a lane owns 4 consecutive columns of each; one
            // v_permlane16_swap per dword trades the odd 16-lane rows' fragment-j half against the even rows'
            // fragment-(j+1) half, after which every lane holds 8 consecutive columns -> 16-byte stores, 64
            // contiguous bytes per output row and instruction (the epilogue is store-ISSUE bound: measured 12.5 us
            // per 256x256 tile with 8-byte stores, profiles/r01g_epilogue_probe.txt).
#if TP_EPI_FULL_LINE
            typedef unsigned u32x4_fl __attribute__((ext_vector_type(4)));
            u32x4_fl pk_fl[2];                          // the packed 64-byte segments of the slice's two fragment pairs
#endif
#pragma unroll
            for (int jj = 0; jj < FNV; jj += 2) {
                const int j0 = vs * FNV + jj, j1 = j0 + 1;
                f32x4 v0 = acc[i][j0], v1 = acc[i][j1];
                if constexpr (LAZY_PAR) {
                    if (flags & TP_LINEAR_LN_FOLD) {
                        const f32x4 c0 = *(const f32x4*)(lds_par + BN * 4 + (lc_lazy + j0 * 16) * 4);
                        const f32x4 c1 = *(const f32x4*)(lds_par + BN * 4 + (lc_lazy + j1 * 16) * 4);
                        v0 = rstd * (v0 - mu * c0); v1 = rstd * (v1 - mu * c1);
                    }
                    v0 += *(const f32x4*)(lds_par + (lc_lazy + j0 * 16) * 4);
                    v1 += *(const f32x4*)(lds_par + (lc_lazy + j1 * 16) * 4);
                } else {
#if TP_EPI_PROBE != 2
                if (flags & TP_LINEAR_LN_FOLD) { v0 = rstd * (v0 - mu * csum_v[j0]); v1 = rstd * (v1 - mu * csum_v[j1]); }
                v0 += bias_v[j0]; v1 += bias_v[j1];
#endif
                }
                if constexpr (TRAIN_EPI) {
                if (flags & TP_LINEAR_GELU_BWD) {          // backward of a GELU layer: dZ = dA * gelu'(Z), Z fp16 [M, ldz]
                    const f16_t* zrow = (const f16_t*)(p.Z + g * p.z_gs) + (long long)(row_ok ? m : 0) * p.ldz + col_base;
                    const f16x4 z0 = *(const f16x4*)(zrow + j0 * 16), z1 = *(const f16x4*)(zrow + j1 * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] *= gelu_erf_grad((float)z0[r]); v1[r] *= gelu_erf_grad((float)z1[r]); }
                }
                if ((flags & TP_LINEAR_SAVE_PRE) && !OUT_F32) {   // training forward: keep the pre-activation (fp16)
                    using O4p = typename Vec<f16_t>::x4;
                    f32x4 c0 = v0, c1 = v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        c0[r] = fminf(fmaxf(c0[r], -65504.f), 65504.f); c1[r] = fminf(fmaxf(c1[r], -65504.f), 65504.f);
                    }
                    f16_t* prow = (f16_t*)(p.C2 + g * p.c2_gs) + (long long)m * p.ldc + col_base;
                    if (row_ok) {
                        *(O4p*)(prow + j0 * 16) = __builtin_convertvector(c0, O4p);
                        *(O4p*)(prow + j1 * 16) = __builtin_convertvector(c1, O4p);
                    }
                }
                }
                if ((flags & TP_LINEAR_GELU) && TP_EPI_PROBE != 2) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2_ev g0 = gelu_erf2(f32x2_ev{v0[r], v0[r + 1]}), g1 = gelu_erf2(f32x2_ev{v1[r], v1[r + 1]});
                        v0[r] = g0[0]; v0[r + 1] = g0[1]; v1[r] = g1[0]; v1[r + 1] = g1[1];
                    }
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                if constexpr (STATS_ONLY) {                             // statistics only: nothing is rounded, nothing stored
                    if (flags & TP_LINEAR_ROW_STATS) {
                        if (jj == 0) pivot = v0[0];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = v0[r] - pivot; s1 += d; s2 = fmaf(d, d, s2); }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = v1[r] - pivot; s1 += d; s2 = fmaf(d, d, s2); }
                    }
                    continue;
                }
                if constexpr (OUT_F32) {
                    const unsigned coff = (unsigned)(((long long)m * p.ldc + col_base - n0 + n0s) * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rsrc_c, (int)(coff + j0 * 64), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rsrc_c, (int)(coff + j1 * 64), 0, 0);
                } else {
                    using O4 = typename Vec<TO>::x4;
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    if constexpr (std::is_same<TO, f16_t>::value && TP_EPI_PROBE != 2) {     // saturate instead of producing inf
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (!TRAIN_EPI) sat_max = sat_track(sat_max, v0[r], v1[r]);
                            // (v_med3_f32 straight: fminf/fmaxf on an MFMA result make hipcc add a canonicalising
                            // v_max_f32 per element — 128 extra VALU instructions per wave and tile)
                            v0[r] = __builtin_amdgcn_fmed3f(v0[r], -65504.f, 65504.f);
                            v1[r] = __builtin_amdgcn_fmed3f(v1[r], -65504.f, 65504.f);
                        }
                    }
                    const O4 o0 = __builtin_convertvector(v0, O4), o1 = __builtin_convertvector(v1, O4);
                    if (flags & TP_LINEAR_ROW_STATS) {                  // stats of the ROUNDED values, fragment order
                        v0 = __builtin_convertvector(o0, f32x4);
                        v1 = __builtin_convertvector(o1, f32x4);
                        if (jj == 0) pivot = v0[0];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = v0[r] - pivot; s1 += d; s2 = fmaf(d, d, s2); }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = v1[r] - pivot; s1 += d; s2 = fmaf(d, d, s2); }
                    }
                    u32x2 a = __builtin_bit_cast(u32x2, o0), b = __builtin_bit_cast(u32x2, o1);
                    const auto sx = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                    // even 16-lane rows now hold columns g*4 .. g*4+7 of fragment j0 (own | upper neighbour's),
                    // odd rows columns (g-1)*4 .. (g-1)*4+7 of fragment j1 (lower neighbour's | own)
                    const int gq = lane >> 4;
#if TP_EPI_PROBE == 1
                    asm volatile("" :: "v"(sx[0]), "v"(sy[0]), "v"(sx[1]), "v"(sy[1]));   // (timing probe: arithmetic, no store)
#elif TP_EPI_FULL_LINE
                    pk_fl[jj >> 1] = u32x4_fl{sx[0], sy[0], sx[1], sy[1]};
                    if (jj == 2) {
                        // WHOLE-LINE stores: the slice's 64 columns are one 128-byte line per row.  Lane l takes (row l / 8 [+ 8 for the second
                        // instruction], 16-byte chunk l % 8): chunk c comes from pair c / 4, lane group gq(c % 4), through ds_bpermute.  One
                        // bpermute serves both instructions: in bpermute h' the source lanes of rows 0..7 offer pair h', those of rows 8..15
                        // pair 1 - h'; a destination lane whose chunk is of pair h' reads row r (first instruction), the others row r + 8.
                        const int c8 = lane & 7, rr = lane >> 3, hp = c8 >> 2;
                        const int gsrc = ((c8 & 1) << 1) | ((c8 >> 1) & 1);
                        const int a0 = (16 * gsrc + rr + (hp ? 8 : 0)) * 4, a1 = (16 * gsrc + rr + (hp ? 0 : 8)) * 4;
                        const bool up = (lane & 8) != 0;
                        u32x4_fl A4, B4;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned s0 = up ? pk_fl[1][d] : pk_fl[0][d], s1 = up ? pk_fl[0][d] : pk_fl[1][d];
                            const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(a0, (int)s0), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(a1, (int)s1);
                            A4[d] = hp ? r1 : r0; B4[d] = hp ? r0 : r1;
                        }
                        const int m_fl = m0 + wm * WM + i * 16 + rr;
                        const int col_fl = n0s + wn * WN + vs * 64 + c8 * 8;
                        const unsigned off_fl = (unsigned)(((long long)m_fl * p.ldc + col_fl) * (long long)sizeof(TO));
                        __builtin_amdgcn_raw_buffer_store_b128(A4, rsrc_c, (int)off_fl, 0, TP_EPI_FULL_LINE == 2 ? 2 : 0);
                        __builtin_amdgcn_raw_buffer_store_b128(B4, rsrc_c, (int)(off_fl + (unsigned)(8 * p.ldc * (int)sizeof(TO))), 0, TP_EPI_FULL_LINE == 2 ? 2 : 0);
                    }
#elif TP_EPI_LANE_ADJ
                    // The 64-byte segment of row r sits in lanes r, r + 16, r + 32, r + 48 (16-byte chunks 0, 2, 1, 3): adjacent lanes
                    // are adjacent ROWS, and the store unit handles that at ~53 cycles per instruction and CU where the same 16 rows x 64 B
                    // with adjacent lanes adjacent in MEMORY take ~16 (tools/probes/store_pattern_probe.hip).  Four ds_bpermute_b32 (the
                    // LDS crossbar, no LDS memory) move (row r, chunk c) to lane 4 r + c.
                    const int src4 = ((((lane & 1) << 1) | ((lane >> 1) & 1)) * 16 + (lane >> 2)) * 4;
                    const unsigned w0 = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)sx[0]), w1 = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)sy[0]);
                    const unsigned w2 = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)sx[1]), w3 = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)sy[1]);
                    const int m_adj = m0 + wm * WM + i * 16 + (lane >> 2);
                    const int col_adj = n0s + wn * WN + j0 * 16 + (lane & 3) * 8;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0, w1, w2, w3}, rsrc_c,
                                                           (int)(unsigned)(((long long)m_adj * p.ldc + col_adj) * (long long)sizeof(TO)), 0, 0);
#else
                    const int col = n0s + wn * WN + (j0 + (gq & 1)) * 16 + (gq >> 1) * 8;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{sx[0], sy[0], sx[1], sy[1]}, rsrc_c,
                                                           (int)(unsigned)(((long long)m * p.ldc + col) * (long long)sizeof(TO)), 0, 0);
#endif
                }
                if constexpr (OUT_F32) {
                    if (flags & TP_LINEAR_ROW_STATS) {                  // (rejected by gemm_launch; kept for completeness)
                        if (jj == 0) pivot = v0[0];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = v0[r] - pivot; s1 += d; s2 = fmaf(d, d, s2); }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = v1[r] - pivot; s1 += d; s2 = fmaf(d, d, s2); }
                    }
                }
            }
            // the lane's 4·FNV values as (mean, M2)
            constexpr float inv_nl = 1.0f / (4 * FNV);
            rs1[i][vs] = fmaf(s1, inv_nl, pivot); rs2[i][vs] = fmaf(-s1 * inv_nl, s1, s2);
        }
        // (pair kernel: one fragment row at a time — interleaved rows cost registers the prefetched parameters of the next tile need)
        if constexpr (LAZY_PAR) __builtin_amdgcn_sched_barrier(0);
    }

    // (not in the training epilogues: their register budget is exhausted — one more live value spills; a training forward
    // is scanned with tp_debug_count_saturated instead)
    if constexpr (std::is_same<TO, f16_t>::value && !STATS_ONLY && !TRAIN_EPI) sat_report(p, sat_max);
    if (flags & TP_LINEAR_ROW_STATS) {
        // reduce over the 4 lane groups that share a row, then over the 64-column slices through LDS
        float* red = (float*)smem;                 // [BN/64][BM][2]
        block_sync_lds();                          // everyone is done with the scratch / K-slab buffers
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int vs = 0; vs < VW; ++vs) {
                // (mean, M2) of equal-sized groups a, b of n values each:  mean = (ma + mb) / 2,
                // M2 = M2a + M2b + (ma - mb)^2 n / 2 — written symmetrically, so both lanes of a pair get the same bits
                float s1 = rs1[i][vs], s2 = rs2[i][vs];
                {
                    const float mb = __shfl_xor(s1, 16), qb = __shfl_xor(s2, 16), d = s1 - mb;
                    s1 = 0.5f * (s1 + mb); s2 = fmaf(d * d, 0.5f * (4 * FNV), s2 + qb);
                }
                {
                    const float mb = __shfl_xor(s1, 32), qb = __shfl_xor(s2, 32), d = s1 - mb;
                    s1 = 0.5f * (s1 + mb); s2 = fmaf(d * d, 0.5f * (8 * FNV), s2 + qb);
                }
                if (lane < 16) {
                    const int rr = wm * WM + i * 16 + lane;
                    red[((wn * VW + vs) * BM + rr) * 2 + 0] = s1;
                    red[((wn * VW + vs) * BM + rr) * 2 + 1] = s2;
                }
            }
        block_sync_lds();
        // one slab per 128 output columns, whatever the tile: the partial-sum tree (lane -> 4 lane groups ->
        // the two 64-column slices of a slab) is identical for every tile / wave shape, so results do not depend
        // on the kernel the batch size selects (bit-exact batch invariance).
        constexpr int SLABS = BN / 128, WPS = 2;
        for (int idx = tid; idx < BM * SLABS; idx += NW * 64) {
            const int rr = idx % BM, sl = idx / BM;
            static_assert(WPS == 2 && 16 * FNV == 64, "a slab is two 64-column slices");
            const float ma = red[((sl * WPS + 0) * BM + rr) * 2], qa = red[((sl * WPS + 0) * BM + rr) * 2 + 1];
            const float mb = red[((sl * WPS + 1) * BM + rr) * 2], qb = red[((sl * WPS + 1) * BM + rr) * 2 + 1];
            const float dm = ma - mb;
            const float s1 = 0.5f * (ma + mb), s2 = fmaf(dm * dm, 32.0f, qa + qb);        // (mean, M2) of the 128 columns
            const int m = m0 + rr;
            if (m < p.M) {
                float* so = p.stats_out + g * p.stats_out_gs + ((long long)(tile_n * SLABS + sl) * p.M + m) * 2;
                *(float2*)so = make_float2(s1, s2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Region attention inside the epilogues of the K and V in-projections (GemmArgs::attn_mode; scale_factor 2).
//
// The rows of the operand are in region-major order (GemmArgs::a_region_s on the first K/V layer), so the s*s = 4 keys of a
// region are 4 consecutive rows: in the accumulator layout (row = lane & 15) they are the 4 lanes of one QUAD, and with
// ONE query per region (builder.py:122-130) attention needs nothing a quad and the two waves of a head (WN = 64, a head =
// 128 columns) do not already hold.  K and V are therefore never written (2 x 302 MB at B = 256) nor read back by an
// attention kernel:
//   K launch  attn_logits_epilogue: K = LayerNorm-folded accumulators in fp32 (not rounded to fp16);  per row the partial
//             dot product with the region's query over the lane's 16 columns, + xor 16, + xor 32 (the wave's 64 columns),
//             + the partner wave of the head through LDS -> logit[head][row] (fp32, scaled), 4 B per row and head.
//   V launch  attn_sum_epilogue:    p = softmax over the quad's 4 logits (every lane computes the same p in the same order),
//             w = p_row * V, summed over the quad with two DPP quad_perm adds; lane k of the quad stores fragment j = k:
//             O[region][cols] fp16, 8 B per lane, FM stores per wave and tile instead of FM * FN / 2 of 16 B.
// The sums run in one fixed order whatever the tile shape (WN = 64 in every kernel), so the result does not depend on
// the kernel a batch size selects.  `lds_q` (persistent kernel): the tile's queries [BM / 4][BN] fp16 staged in LDS;
// `lds_par`: bias[BN] | colsum[BN] staged by the caller, and for the V launch the tile's logits [BN / 128][BM] at +LG_OFF.
// ---------------------------------------------------------------------------------------------------------------------
// Q_LDS: the tile's queries are staged in LDS (`lds_q`); false: read from GemmArgs::attn_q by the epilogue (tp_gemm_pair.hip)
template <int BM, int BN, int WM, int WN, bool LDS_PARAMS, bool Q_LDS = LDS_PARAMS>
__device__ __forceinline__ void attn_logits_epilogue(f32x4 (&acc)[WM / 16][WN / 16], const GemmArgs& p, const int g,
                                                     const int m0, const int n0, const int wm, const int wn,
                                                     const int lane, const int tid, const float2 (&mean_rstd)[WM / 16],
                                                     char* smem, const char* lds_par = nullptr, const char* lds_q = nullptr) {
    constexpr int NWM = BM / WM, NWN = BN / WN, NW = NWM * NWN;
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(WN == 64 && BN % 128 == 0, "a head is two 64-column waves");
    const int cg = lane >> 4, r16 = lane & 15;
    const int col_base = n0 + wn * WN + cg * 4;
    f32x4 bias_v[FN], csum_v[FN];
    if constexpr (LDS_PARAMS) {
        // (read from LDS at the point of use: 32 registers less across the epilogue of the persistent kernels)
    } else {
        const float* __restrict__ bias = p.bias + g * p.bias_gs;
        const float* __restrict__ colsum = p.colsum + g * p.colsum_gs;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bias_v[j] = *(const f32x4*)(bias + col_base + j * 16);
            csum_v[j] = *(const f32x4*)(colsum + col_base + j * 16);
        }
    }
    // Q_LDS = false with staged parameters (tp_gemm_pair.hip): every query fragment of the wave's rows is fetched up front — one
    // memory round trip instead of one per fragment row (the fragment registers of the K loop are free here)
    constexpr bool Q_AHEAD = LDS_PARAMS && !Q_LDS;
    f16x4 qv[Q_AHEAD ? FM : 1][Q_AHEAD ? FN : 1];
    if constexpr (Q_AHEAD) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            int m = m0 + wm * WM + i * 16 + r16;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int j = 0; j < FN; ++j)
                qv[i][j] = *(const f16x4*)(p.attn_q + (long long)(m >> 2) * p.attn_ldq_bytes + (col_base + j * 16) * 2);
        }
    }
    float part[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int rr = wm * WM + i * 16 + r16;             // row inside the tile
        float mu = mean_rstd[i].x, rstd = mean_rstd[i].y;
        if constexpr (LDS_PARAMS) {                        // (read per fragment row: 14 registers less across the epilogue)
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t mr = *(const f32x2_t*)(lds_par + 2 * BN * 4 + rr * 8);
            mu = mr[0]; rstd = mr[1];
        }
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            f16x4 q;
            if constexpr (Q_AHEAD) {
                q = qv[i][j];
            } else if constexpr (Q_LDS) {
                q = *(const f16x4*)(lds_q + (rr >> 2) * (BN * 2) + (wn * WN + j * 16 + cg * 4) * 2);
            } else {
                int m = m0 + rr;
                m = m < p.M ? m : p.M - 1;
                q = *(const f16x4*)(p.attn_q + (long long)(m >> 2) * p.attn_ldq_bytes + (col_base + j * 16) * 2);
            }
            f32x4 bj, cj;
            if constexpr (LDS_PARAMS) {
                bj = *(const f32x4*)(lds_par + (wn * WN + cg * 4 + j * 16) * 4);
                cj = *(const f32x4*)(lds_par + BN * 4 + (wn * WN + cg * 4 + j * 16) * 4);
            } else { bj = bias_v[j]; cj = csum_v[j]; }
            // (decoupled: the raw accumulators — rstd is applied by the V launch, the bias term is softmax-invariant)
            const f32x4 v = p.attn_decoupled ? acc[i][j] : rstd * (acc[i][j] - mu * cj) + bj;
#pragma unroll
            for (int r = 0; r < 4; ++r) d = fmaf(v[r], (float)q[r], d);
        }
        d += __shfl_xor(d, 16);
        d += __shfl_xor(d, 32);
        part[i] = d;
        __builtin_amdgcn_sched_barrier(0);                 // one fragment row at a time: the persistent kernels have no registers to spare
    }
    float* red = (float*)smem;                             // [NWN][BM]
    block_sync_lds();                                      // everyone is done with the scratch / K-slab buffers
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < FM; ++i) red[wn * BM + wm * WM + i * 16 + lane] = part[i];
    }
    block_sync_lds();
    constexpr int HEADS = BN / 128;
    for (int idx = tid; idx < BM * HEADS; idx += NW * 64) {
        const int rr = idx % BM, hl = idx / BM;
        const float lg = (red[(2 * hl) * BM + rr] + red[(2 * hl + 1) * BM + rr]) * p.attn_scale;
        const int m = m0 + rr;
        if (m < p.M) p.attn_logits[(long long)(n0 / 128 + hl) * p.M + m] = lg;
    }
}

template <int BM, int BN, int WM, int WN, bool LDS_PARAMS, int LG_OFF = 0>
__device__ __forceinline__ void attn_sum_epilogue(f32x4 (&acc)[WM / 16][WN / 16], const GemmArgs& p, const int g,
                                                  const int m0, const int n0, const int wm, const int wn,
                                                  const int lane, const float2 (&mean_rstd)[WM / 16],
                                                  const char* lds_par = nullptr) {
    constexpr int FM = WM / 16, FN = WN / 16;
    static_assert(WN == 64 && FN == 4, "lane k of a quad stores fragment k");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int cg = lane >> 4, r16 = lane & 15, kq = lane & 3;
    const int col_base = n0 + wn * WN + cg * 4;
    const int hl = (wn * WN) / 128;                        // head inside the tile
    float sat_max = 0.f;
    // O has M / 4 rows: the hardware range check drops the stores of rows past the end (uniform store count per tile)
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.C + g * p.c_gs), 0, (int)(unsigned)((long long)(p.M >> 2) * p.ldc * 2), 0x00020000);
    f32x4 bias_v[FN], csum_v[FN];
    if constexpr (LDS_PARAMS) {
        // (read from LDS at the point of use: 32 registers less across the epilogue of the persistent kernels)
    } else {
        const float* __restrict__ bias = p.bias + g * p.bias_gs;
        const float* __restrict__ colsum = p.colsum + g * p.colsum_gs;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bias_v[j] = *(const f32x4*)(bias + col_base + j * 16);
            csum_v[j] = *(const f32x4*)(colsum + col_base + j * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int rr = wm * WM + i * 16 + r16;
        const int m = m0 + rr;
        f32x4 lg;                                          // the logits of the quad's 4 rows, this wave's head
        if constexpr (LDS_PARAMS) {
            lg = *(const f32x4*)(lds_par + LG_OFF + (hl * BM + (rr & ~3)) * 4);
        } else {
            const int mc = m < p.M ? m : p.M - 4;          // (M % 4 == 0: a quad is valid or past the end as a whole)
            lg = *(const f32x4*)(p.attn_logits + (long long)(n0 / 128 + hl) * p.M + (mc & ~3));
        }
        float mu = mean_rstd[i].x, rstd = mean_rstd[i].y;
        if constexpr (LDS_PARAMS) {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t mr = *(const f32x2_t*)(lds_par + 2 * BN * 4 + rr * 8);
            mu = mr[0]; rstd = mr[1];
        }
        if (p.attn_decoupled) {
            // the mean slot carries the K row's rstd (GemmArgs::attn_decoupled): lane kq of the quad holds row kq's — fetch the
            // other three by DPP quad broadcasts; every lane then scales the same four raw logits in the same order
            const int rk = __builtin_bit_cast(int, mu);
            lg[0] *= __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(rk, 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
            lg[1] *= __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(rk, 0x55, 0xF, 0xF, true));   // [1,1,1,1]
            lg[2] *= __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(rk, 0xAA, 0xF, 0xF, true));   // [2,2,2,2]
            lg[3] *= __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(rk, 0xFF, 0xF, 0xF, true));   // [3,3,3,3]
            mu = 0.f;
        }
        const float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        const float e0 = __expf(lg[0] - mx), e1 = __expf(lg[1] - mx), e2 = __expf(lg[2] - mx), e3 = __expf(lg[3] - mx);
        const float den = ((e0 + e1) + e2) + e3;
        const float pr = (kq == 0 ? e0 : kq == 1 ? e1 : kq == 2 ? e2 : e3) / den;
        f32x4 mine = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            f32x4 bj, cj;
            if constexpr (LDS_PARAMS) {
                bj = *(const f32x4*)(lds_par + (wn * WN + cg * 4 + j * 16) * 4);
                cj = *(const f32x4*)(lds_par + BN * 4 + (wn * WN + cg * 4 + j * 16) * 4);
            } else { bj = bias_v[j]; cj = csum_v[j]; }
            f32x4 w = pr * (rstd * (acc[i][j] - mu * cj) + bj);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = w[r];
                x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                w[r] = x;
            }
            if (kq == j) mine = w;
        }
        sat_max = sat_track(sat_track(sat_max, mine[0], mine[1]), mine[2], mine[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = __builtin_amdgcn_fmed3f(mine[r], -65504.f, 65504.f);
        const f16x4 o = __builtin_convertvector(mine, f16x4);
        const long long off = ((long long)(m >> 2) * p.ldc + n0 + wn * WN + kq * 16 + cg * 4) * 2;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsrc_o, (int)(unsigned)off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // one fragment row at a time (register budget of the persistent kernels)
    }
    sat_report(p, sat_max);
}

}  // namespace tp
