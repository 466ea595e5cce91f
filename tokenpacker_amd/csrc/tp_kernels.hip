// tp_kernels.hip — the HBM-bound pieces of the TokenPacker path for gfx950:
//   * point_queries_kernel   : fp32 bilinear downsample -> coarse point queries  (builder.py:117-118)
//   * region_attention_kernel: region gather + 8-head softmax(q·K^T/sqrt(d))·V with ONE query per
//                              region (builder.py:96-105, 122-130)
//   * pack_* kernels          : one-time weight preparation (LayerNorm fold, fp32 biases)
// All loads/stores are 16 B per lane, 64-lane coalesced (1 KiB per wave instruction).
#include "tp_internal.h"

namespace tp {

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&f)[8]) {
    using X8 = typename Vec<T>::x8;
    const X8 v = *(const X8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&f)[8]) {
    using X8 = typename Vec<T>::x8;
    X8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
    *(X8*)p = v;
}

// ---------------------------------------------------------------------------------------------------
// Coarse point queries.  F.interpolate(bilinear, align_corners=False) from g x g to G x G with the
// integer ratio s = g/G: src = (i + 0.5)*s - 0.5, so for even s the two taps are rows s*i+s/2-1 and
// s*i+s/2 with weights 0.5/0.5, for odd s the single tap s*i+(s-1)/2 with weight 1 (SURVEY.md §8a:
// s=2 2x2 mean, s=3 centre pixel, s=4 inner 2x2 mean).  Arithmetic in fp32 in PyTorch's tap order,
// result rounded once to T (builder.py:117-118).  One thread = 8 channels of one query.
template <typename T>
__global__ void __launch_bounds__(256)
point_queries_kernel(const T* __restrict__ x, long long sb, long long st, f16_t* __restrict__ q0,
                     int B, int g, int s, int C) {
    const int G = g / s, M = G * G, vecs = C / 8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * M * vecs) return;
    const int cv = (int)(gid % vecs);
    const long long qm = gid / vecs;
    const int m = (int)(qm % M), b = (int)(qm / M);
    const int i = m / G, j = m % G;
    const T* xb = x + b * sb + cv * 8;
    float out[8];
    if (s & 1) {
        const int r = s * i + (s - 1) / 2, c = s * j + (s - 1) / 2;
        load8(xb + (long long)(r * g + c) * st, out);
    } else {
        const int r = s * i + s / 2 - 1, c = s * j + s / 2 - 1;
        float a00[8], a01[8], a10[8], a11[8];
        load8(xb + (long long)(r * g + c) * st, a00);
        load8(xb + (long long)(r * g + c + 1) * st, a01);
        load8(xb + (long long)((r + 1) * g + c) * st, a10);
        load8(xb + (long long)((r + 1) * g + c + 1) * st, a11);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            out[e] = 0.5f * (0.5f * a00[e] + 0.5f * a01[e]) + 0.5f * (0.5f * a10[e] + 0.5f * a11[e]);
    }
    // round once to the INPUT dtype (the reference's `.to(x.dtype)`), then widen exactly to the fp16
    // activation type (saturating: a bf16 value can exceed the fp16 range)
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = fminf(fmaxf((float)(T)out[e], -65504.f), 65504.f);
    store8(q0 + qm * C + cv * 8, out);
}

int point_queries_launch(int dtype, const void* x, const int64_t st[3], void* q0, int B, int grid,
                         int s, hipStream_t stream) {   // q0 is fp16
    const int G = grid / s, M = G * G;
    const long long total = (long long)B * M * (kEmbed / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(point_queries_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream,
                           (const bf16_t*)x, (long long)st[0], (long long)st[1], (f16_t*)q0, B, grid, s, kEmbed);
    else
        hipLaunchKernelGGL(point_queries_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream,
                           (const f16_t*)x, (long long)st[0], (long long)st[1], (f16_t*)q0, B, grid, s, kEmbed);
    return check_launch("point_queries_kernel");
}

// ---------------------------------------------------------------------------------------------------
// Region-to-point attention.  One wavefront per coarse query (B*M of them), four per workgroup.
// E = 1024 = 64 lanes x 16 elements: lane l owns elements [8l, 8l+8) (head l>>4) and
// [512+8l, 512+8l+8) (head 4 + (l>>4)), so every global access is a fully coalesced 1 KiB wave
// transaction and each head's 128-wide dot product is a 16-lane butterfly.  The s*s keys of the
// region (token (i*s+a)*g + j*s+b — the reference's divide_feature order, builder.py:96-105; the
// order is irrelevant to softmax·V) are streamed in groups of 4 straight into registers: every K/V
// row is used by exactly one query, so staging it in LDS would be pure overhead (guide: operand
// streamed once and not shared -> load straight to VGPRs).  Softmax is fp32, online across groups,
// which makes the kernel valid for any s dividing the grid (s*s from 1 to 576 keys).
template <typename T>
__global__ void __launch_bounds__(256)
region_attention_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                        T* __restrict__ o, int B, int g, int s, float scale, const float* __restrict__ mask, int mask_mode,
                        int region_major) {
    constexpr int E = kEmbed;
    const int lane = threadIdx.x & 63;
    const int G = g / s, M = G * G, N = g * g, S2 = s * s;
    const long long qi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= (long long)B * M) return;
    const int b = (int)(qi / M), m = (int)(qi % M);
    const int i = m / G, j = m % G;
    const int ea = lane * 8, eb = 512 + lane * 8;

    float qa[8], qb[8];
    load8(q + qi * E + ea, qa);
    load8(q + qi * E + eb, qb);
#pragma unroll
    for (int e = 0; e < 8; ++e) { qa[e] *= scale; qb[e] *= scale; }

    const T* kb = k + (long long)b * N * E;
    const T* vb = v + (long long)b * N * E;
    // additive attn_mask of nn.MultiheadAttention (builder.py:107,130): [S2] for every (image, region, head), or
    // [(M B) 8, S2] with the reference's batch index (region m) * B + image b  (divide_feature's order, builder.py:96-105)
    const float* mrow_a = nullptr; const float* mrow_b = nullptr;
    if (mask_mode == 1) { mrow_a = mask; mrow_b = mask; }
    else if (mask_mode == 2) {
        const long long bi = (long long)m * B + b;
        mrow_a = mask + (bi * kHeads + (lane >> 4)) * S2;
        mrow_b = mask + (bi * kHeads + 4 + (lane >> 4)) * S2;
    }

    float run_max_a = -INFINITY, run_max_b = -INFINITY, den_a = 0.f, den_b = 0.f;
    float acc_a[8], acc_b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc_a[e] = 0.f; acc_b[e] = 0.f; }

    constexpr int KU = 4;
    for (int k0 = 0; k0 < S2; k0 += KU) {
        float la[KU], lb[KU];
        long long row[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kk = (k0 + u < S2) ? k0 + u : S2 - 1;     // clamp: masked below
            const int a = kk / s, c = kk - a * s;
            row[u] = region_major ? (long long)(m * S2 + kk) * E : (long long)((i * s + a) * g + j * s + c) * E;
            float ka[8], kbv[8];
            load8(kb + row[u] + ea, ka);
            load8(kb + row[u] + eb, kbv);
            float da = 0.f, db = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { da = fmaf(qa[e], ka[e], da); db = fmaf(qb[e], kbv[e], db); }
            la[u] = da; lb[u] = db;
        }
        // 16-lane butterflies: every lane of a head group ends with the full 128-wide dot product
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                la[u] += __shfl_xor(la[u], off);
                lb[u] += __shfl_xor(lb[u], off);
            }
        }
        if (mrow_a) {
#pragma unroll
            for (int u = 0; u < KU; ++u)
                if (k0 + u < S2) { la[u] += mrow_a[k0 + u]; lb[u] += mrow_b[k0 + u]; }
        }
        float gmax_a = run_max_a, gmax_b = run_max_b;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (k0 + u < S2) { gmax_a = fmaxf(gmax_a, la[u]); gmax_b = fmaxf(gmax_b, lb[u]); }
        }
        const float resc_a = __expf(run_max_a - gmax_a), resc_b = __expf(run_max_b - gmax_b);  // exp(-inf)=0 first time
        den_a *= resc_a; den_b *= resc_b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc_a[e] *= resc_a; acc_b[e] *= resc_b; }
        run_max_a = gmax_a; run_max_b = gmax_b;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (k0 + u < S2) {
                const float pa = __expf(la[u] - run_max_a), pb = __expf(lb[u] - run_max_b);
                den_a += pa; den_b += pb;
                float va[8], vbv[8];
                load8(vb + row[u] + ea, va);
                load8(vb + row[u] + eb, vbv);
#pragma unroll
                for (int e = 0; e < 8; ++e) { acc_a[e] = fmaf(pa, va[e], acc_a[e]); acc_b[e] = fmaf(pb, vbv[e], acc_b[e]); }
            }
        }
    }
    const float inv_a = 1.0f / den_a, inv_b = 1.0f / den_b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc_a[e] *= inv_a; acc_b[e] *= inv_b; }
    store8(o + qi * E + ea, acc_a);
    store8(o + qi * E + eb, acc_b);
}

int region_attention_launch(const void* q, const void* k, const void* v, void* o, int B,
                            int grid, int s, hipStream_t stream, const float* mask, int mask_mode, int region_major) {
    const int G = grid / s, M = G * G;
    const long long nq = (long long)B * M;
    const unsigned blocks = (unsigned)((nq + 3) / 4);
    const float scale = 0.08838834764831845f;   // 1/sqrt(128): q scaling of F.multi_head_attention_forward
    hipLaunchKernelGGL(region_attention_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream,
                       (const f16_t*)q, (const f16_t*)k, (const f16_t*)v, (f16_t*)o, B, grid, s, scale, mask, mask_mode, region_major);
    return check_launch("region_attention_kernel");
}

// ---------------------------------------------------------------------------------------------------
// Region-to-point attention with the K/V in-projections ABSORBED into the query side (scale_factor >= 3).
//
// With ONE query per region the in-projections of nn.MultiheadAttention (torch/nn/functional.py:5854-5860; three
// separate linears because q, k, v are different tensors) need not be applied to the s*s keys and values of every
// region.  With n_t = LayerNorm-normalised row t of H2 (mean / rstd applied, the affine folded into W' = W·diag(gamma),
// b' = W·beta + b as everywhere else in this library):
//     K_t = W'k n_t + b'k        =>   Q_h · K_t,h = (W'k_h^T Q_h) · n_t + Q_h · b'k_h  = qt_h · n_t + const(t)
//     V_t = W'v n_t + b'v        =>   sum_t p_t V_t,h = W'v_h (sum_t p_t n_t) + b'v_h  = W'v_h u_h + b'v_h
// (const(t) is the same for every key of the region: softmax ignores it; sum_t p_t = 1.)  So the 2 x [B 576, 1024] x
// [1024, 1024] in-projection GEMMs (2.4 GFLOP / image, 604 MB of K/V written and read again at B = 256) become
//   qt = per-head Q_h · W'k_h   [B M, 8, 1024]   (a K = 128 GEMM over the B·M queries, 1/s^2 of the FLOPs)
//   this kernel: logits from qt and the rows of H2k normalised ON LOAD, softmax, u_h = sum_t p_t n^v_t
//   O_h = u_h · W'v_h^T + b'v_h                  (a per-head GEMM over the B·M queries)
// Same function, same parity gates; the roofline is still priced on the un-absorbed FLOP count.
//
// One wavefront per query, four per workgroup; lane l owns elements [8l, 8l+8) and [512+8l, 512+8l+8) of every
// 1024-vector (two coalesced 1-KiB wave transactions per row).  Phase A: the 8 per-head dot products of every key
// are reduced over the wave with a halving exchange (10 shuffles per key instead of 48) and parked in 2 KiB of LDS;
// softmax runs there; phase B re-walks the region's tokens on the V side, accumulating u in 128 fp32 registers.
// HBM-bound: reads 2 s^2 rows of H2 + 16 KiB of qt, writes 16 KiB of u per query.
//
// RAW form (the fused LayerNorm chain under the absorbed schedule, TP_TUNE_FUSE_KV_LN): H2 = Hkv·W2^T + b2 is computed for
// its row statistics only and never stored; the kernel walks the rows of Hkv itself (row stride `ld`) and the second layer
// rides in the pre-multiplied weights Wc = W'·W2, d = W'·b2, c = rowsum(W') of the fused chain:
//     Q_h·K_t,h = rstd_t (qt_h·hkv_t + alpha_h - mu_t beta_h) + const,   qt_h = Wc_k,h^T Q_h,  alpha_h = Q_h·d_k,h,  beta_h = Q_h·c_k,h
//     sum_t p_t V_t,h = a_h (Wc_v,h u_h + d_v,h - (e_h / a_h) c_v,h) + b'v_h,
//         w_t = p_t rstd_t,  a_h = sum_t w_t,  e_h = sum_t w_t mu_t,  u_h = (sum_t w_t hkv^v_t) / a_h
// i.e. the per-head V GEMM behind it is a LayerNorm-fold GEMM with (mean, rstd) := (e_h / a_h, a_h), written to `mr_u`
// ([8 heads][B M][2]).  No per-element normalisation on load: two VALU operations less per element and key.
// Waves per SIMD the absorbed attention kernel is compiled for.  2: 190 VGPRs, no scratch.  3 (-DTP_ABSORB_WAVES=3, `make variant
// VSRC=tp_kernels`): 168 VGPRs + 29 spilled — measured 0.52 -> 0.645 ms for the attention stage at B = 256, s = 3 (0.39 -> 0.48 at s = 4):
// the kernel is HBM-bound at 4.6-4.9 TB/s and the spills cost more than the third wave's loads in flight buy (profiles/r05k_absorb_waves_ab.txt).
#ifndef TP_ABSORB_WAVES
#define TP_ABSORB_WAVES 2
#endif
constexpr int kAbsorbMaxKeys = 64;          // s*s <= 64 (s <= 8): logits of a region live in LDS

// QT32 (the library's own s >= 3 schedule): qt arrives in fp32 — the per-head query GEMM's accumulators, not rounded.  (Round 4 also wrote
// u as hi | lo fp16 halves, u_ld = 2 E; round 5 measured that residual worth nothing on the worst seeds and 0.12 ms per forward: gone.)
template <bool RAW, bool QT32 = false>
__global__ void __launch_bounds__(256, TP_ABSORB_WAVES)
region_attention_absorbed_kernel(const void* __restrict__ qt_, const f16_t* __restrict__ h2k, const f16_t* __restrict__ h2v,
                                 const float* __restrict__ mr_k, const float* __restrict__ mr_v, f16_t* __restrict__ u,
                                 int B, int g, int s, float scale, const float* __restrict__ mask, int mask_mode,
                                 int ld, const f16_t* __restrict__ q, const float* __restrict__ d_k,
                                 const float* __restrict__ c_k, float* __restrict__ mr_u, const int u_ld) {
    constexpr int E = kEmbed, H = kHeads;
    __shared__ float logit_lds[4][kAbsorbMaxKeys * H];
    __shared__ float ab_lds[4][2 * H];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int G = g / s, M = G * G, N = g * g, S2 = s * s;
    const long long qi = (long long)blockIdx.x * 4 + wv;
    if (qi >= (long long)B * M) return;                       // (no workgroup barrier below: waves are independent)
    const int b = (int)(qi / M), m = (int)(qi % M);
    const int i = m / G, j = m % G;
    const int ea = lane * 8, eb = 512 + lane * 8;
    float* lg = logit_lds[wv];
    auto token_row = [&](int kk) -> long long {
        const int a = kk / s, c = kk - a * s;
        return (long long)b * N + (long long)((i * s + a) * g + j * s + c);
    };
    // Rows are fetched ONE KEY AHEAD (packed, 8 VGPRs + the row's statistics) so that two rows per wave are in flight:
    // at 2 waves / SIMD (the 128 fp32 accumulators of phase B) a CU would otherwise keep 16 KiB outstanding (3.9 -> 4.65
    // TB/s).  Measured and NOT kept (r02g, same box): two keys ahead = the same; both phases as two passes of four heads
    // (133 VGPRs -> 3 waves / SIMD, or 128 with 24 B of scratch -> 4) = 11 % SLOWER — the rows of the second pass cost
    // more out of L2 than the extra waves hide.
    struct Row { f16x8 a, b; float2 st; };
    auto fetch = [&](const f16_t* __restrict__ base, const float* __restrict__ mr, int t) -> Row {
        const long long row = token_row(t < S2 ? t : S2 - 1);
        Row r;
        r.st = *(const float2*)(mr + row * 2);                  // (mean, rstd): the same address for all lanes
        r.a = *(const f16x8*)(base + row * ld + ea);
        r.b = *(const f16x8*)(base + row * ld + eb);
        return r;
    };
    auto normalise = [&](const Row& r, float (&na)[8], float (&nb)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (RAW) { na[e] = (float)r.a[e]; nb[e] = (float)r.b[e]; }
            else { na[e] = ((float)r.a[e] - r.st.x) * r.st.y; nb[e] = ((float)r.b[e] - r.st.x) * r.st.y; }
        }
    };
    {   // ---- phase A: logits ---------------------------------------------------------------------------------
        using QV = typename std::conditional<QT32, f32x8_t, f16x8>::type;
        const typename std::conditional<QT32, float, f16_t>::type* qt = (decltype(qt))qt_;
        QV qa[H], qb[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            qa[h] = *(const QV*)(qt + (qi * H + h) * E + ea);
            qb[h] = *(const QV*)(qt + (qi * H + h) * E + eb);
        }
        Row cur = fetch(h2k, mr_k, 0);
        // (RAW: the per-head scalars are computed while qt and the first row are in flight)
        if constexpr (RAW) {
            // alpha_h = Q_h·d_k,h, beta_h = Q_h·c_k,h: lane l holds elements of head l / 16 (low half) and 4 + l / 16 (high half)
            float qa8[8], qb8[8];
            load8(q + qi * E + ea, qa8);
            load8(q + qi * E + eb, qb8);
            float al = 0.f, ah = 0.f, bl = 0.f, bh = 0.f;
    #pragma unroll
            for (int e = 0; e < 8; ++e) {
                al = fmaf(qa8[e], d_k[ea + e], al); bl = fmaf(qa8[e], c_k[ea + e], bl);
                ah = fmaf(qb8[e], d_k[eb + e], ah); bh = fmaf(qb8[e], c_k[eb + e], bh);
            }
    #pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                al += __shfl_xor(al, off); bl += __shfl_xor(bl, off); ah += __shfl_xor(ah, off); bh += __shfl_xor(bh, off);
            }
            if ((lane & 15) == 0) {
                const int hg = lane >> 4;
                ab_lds[wv][hg] = al; ab_lds[wv][4 + hg] = ah; ab_lds[wv][H + hg] = bl; ab_lds[wv][H + 4 + hg] = bh;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        for (int t = 0; t < S2; ++t) {
            const Row nxt = fetch(h2k, mr_k, t + 1);
            float na[8], nb[8], v[H];
            normalise(cur, na, nb);
            const float2 st_t = cur.st;
            cur = nxt;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { acc = fmaf(na[e], (float)qa[h][e], acc); acc = fmaf(nb[e], (float)qb[h][e], acc); }
                v[h] = acc;
            }
            // halving exchange: after the three steps lane l holds the sum over its 8-lane group's partners of head
            // hsel(l) = 4 bit5 + 2 bit4 + bit3; three plain butterflies finish the 64-lane sum
            float w4[4], w2[2], y;
            {
                const bool hi = lane & 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float send = hi ? v[q] : v[q + 4], keep = hi ? v[q + 4] : v[q]; w4[q] = keep + __shfl_xor(send, 32); }
            }
            {
                const bool hi = lane & 16;
#pragma unroll
                for (int q = 0; q < 2; ++q) { const float send = hi ? w4[q] : w4[q + 2], keep = hi ? w4[q + 2] : w4[q]; w2[q] = keep + __shfl_xor(send, 16); }
            }
            {
                const bool hi = lane & 8;
                const float send = hi ? w2[0] : w2[1], keep = hi ? w2[1] : w2[0];
                y = keep + __shfl_xor(send, 8);
            }
            y += __shfl_xor(y, 4); y += __shfl_xor(y, 2); y += __shfl_xor(y, 1);
            if ((lane & 7) == 0) {
                const int hsel = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
                float lgt = y * scale;
                if constexpr (RAW)                                                         // cur was advanced: its stats are in `st_t`
                    lgt = st_t.y * (y + ab_lds[wv][hsel] - st_t.x * ab_lds[wv][H + hsel]) * scale;
                if (mask_mode == 1) lgt += mask[t];                                       // attn_mask, see region_attention_kernel
                else if (mask_mode == 2) lgt += mask[(((long long)m * B + b) * H + hsel) * S2 + t];
                lg[t * H + hsel] = lgt;
            }
        }
    }
    // ---- softmax over the region's keys, one lane per head (LDS traffic of ONE wave is ordered; the asm keeps hipcc
    // from moving the accesses across) ---------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < H) {
        float mx = -INFINITY;
        for (int t = 0; t < S2; ++t) mx = fmaxf(mx, lg[t * H + lane]);
        float den = 0.f;
        for (int t = 0; t < S2; ++t) { const float p = __expf(lg[t * H + lane] - mx); den += p; lg[t * H + lane] = p; }
        const float inv = 1.0f / den;
        for (int t = 0; t < S2; ++t) lg[t * H + lane] *= inv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- phase B: u_h = sum_t p_t,h n^v_t ---------------------------------------------------------------------------
    float ua[H][8], ub[H][8];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ua[h][e] = 0.f; ub[h][e] = 0.f; }
    float a_h[H], e_h[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { a_h[h] = 0.f; e_h[h] = 0.f; }
    Row cur = fetch(h2v, mr_v, 0);
    for (int t = 0; t < S2; ++t) {
        const Row nxt = fetch(h2v, mr_v, t + 1);
        float na[8], nb[8];
        normalise(cur, na, nb);
        const float2 st_t = cur.st;
        cur = nxt;
        const f32x4 p0 = *(const f32x4*)(lg + t * H), p1 = *(const f32x4*)(lg + t * H + 4);     // broadcast reads
        float p[H] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
        if constexpr (RAW) {
#pragma unroll
            for (int h = 0; h < H; ++h) { p[h] *= st_t.y; a_h[h] += p[h]; e_h[h] = fmaf(p[h], st_t.x, e_h[h]); }
        }
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ua[h][e] = fmaf(p[h], na[e], ua[h][e]); ub[h][e] = fmaf(p[h], nb[e], ub[h][e]); }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        if constexpr (RAW) {
            const float inv = 1.0f / a_h[h];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ua[h][e] *= inv; ub[h][e] *= inv; }
            if (lane == h) *(float2*)(mr_u + ((long long)h * B * M + qi) * 2) = make_float2(e_h[h] * inv, a_h[h]);
        }
        store8(u + (qi * H + h) * u_ld + ea, ua[h]);
        store8(u + (qi * H + h) * u_ld + eb, ub[h]);
    }
}

int region_attention_absorbed_launch(const void* qt, const void* h2k, const void* h2v, const float* mr_k, const float* mr_v,
                                     void* u, int B, int grid, int s, hipStream_t stream, const float* mask, int mask_mode,
                                     int ld, const void* q, const float* d_k, const float* c_k, float* mr_u, bool u_split) {
    if (s * s > kAbsorbMaxKeys) { set_error("absorbed region attention: s*s = %d keys > %d", s * s, kAbsorbMaxKeys); return TP_ERR_INVALID_ARG; }
    const int G = grid / s, M = G * G;
    const long long nq = (long long)B * M;
    const unsigned blocks = (unsigned)((nq + 3) / 4);
    if (q && u_split)   // RAW, the default of s >= 3: qt in fp32 (u is one fp16 value per element: tp_api.hip)
        hipLaunchKernelGGL((region_attention_absorbed_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, qt,
                           (const f16_t*)h2k, (const f16_t*)h2v, mr_k, mr_v, (f16_t*)u, B, grid, s, 0.08838834764831845f, mask,
                           mask_mode, ld, (const f16_t*)q, d_k, c_k, mr_u, kEmbed);
    else if (q)         // RAW: rows of Hkv, the second K/V layer in the pre-multiplied weights
        hipLaunchKernelGGL(region_attention_absorbed_kernel<true>, dim3(blocks), dim3(256), 0, stream, qt,
                           (const f16_t*)h2k, (const f16_t*)h2v, mr_k, mr_v, (f16_t*)u, B, grid, s, 0.08838834764831845f, mask,
                           mask_mode, ld, (const f16_t*)q, d_k, c_k, mr_u, kEmbed);
    else
        hipLaunchKernelGGL(region_attention_absorbed_kernel<false>, dim3(blocks), dim3(256), 0, stream, qt,
                           (const f16_t*)h2k, (const f16_t*)h2v, mr_k, mr_v, (f16_t*)u, B, grid, s, 0.08838834764831845f, mask,
                           mask_mode, kEmbed, nullptr, nullptr, nullptr, nullptr, kEmbed);
    return check_launch("region_attention_absorbed_kernel");
}

// w_qt[h][j][d] = W'k[h*128 + d][j]: the per-head transposes the qt GEMM reads as its [N = 1024, K = 128] operand
__global__ void __launch_bounds__(256)
pack_head_transpose_kernel(const f16_t* __restrict__ src, f16_t* __restrict__ dst) {
    __shared__ f16_t tile[32][33];
    const int h = blockIdx.z, j0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(long long)(h * kHeadDim + d0 + r) * kEmbed + j0 + tx];   // [d][j]
    __syncthreads();
    for (int r = ty; r < 32; r += 8) dst[((long long)h * kEmbed + j0 + r) * kHeadDim + d0 + tx] = tile[tx][r];    // [j][d]
}

int pack_head_transpose_launch(const void* w_f16, void* dst_f16, hipStream_t stream) {
    hipLaunchKernelGGL(pack_head_transpose_kernel, dim3(kEmbed / 32, kHeadDim / 32, kHeads), dim3(256), 0, stream,
                       (const f16_t*)w_f16, (f16_t*)dst_f16);
    return check_launch("pack_head_transpose_kernel");
}

// ---------------------------------------------------------------------------------------------------
// Test hook: keep `blocks` CUs busy for about `usec` microseconds (100 KiB of LDS per workgroup, so none of the
// path's 140-KiB workgroups fits beside it) — stands in for a collective's kernels on another stream when the
// tile queue of the persistent GEMM is exercised (tools/hog_bench.py).
__global__ void __launch_bounds__(256)
occupy_cus_kernel(long long cycles, int* sink) {
    extern __shared__ int hog_lds[];
    const long long t0 = __builtin_readcyclecounter();
    int acc = 0;
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) { hog_lds[threadIdx.x] = acc; acc += hog_lds[(threadIdx.x + 1) & 255]; }
    if (acc == 0x7fffffff) *sink = acc;
}

int occupy_cus_launch(int blocks, int usec, int* sink, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                100 * 1024) != hipSuccess) {
            set_error("occupy_cus: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            return TP_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(blocks), dim3(256), 100 * 1024, stream, (long long)usec * 2000, sink);   // s_memtime ticks at the shader clock (~2 GHz)
    return check_launch("occupy_cus_kernel");
}

// ---------------------------------------------------------------------------------------------------
// TokenPacker-HD token assembly (llava_arch.py:140-154).  One workgroup per output row of D 16-bit elements,
// 16 B per lane.  Row r of an image with an h x w grid: the first h*w*(M+1) rows are h*w segments of M crop
// tokens + 1 separator (',' unless the crop ends its grid row, then '\n'); with more than one crop, M rows of
// the global view and a final '\n' follow.  The per-image plan (<= 64 images per launch) travels by value.
struct HdPlan { tp_hd_image img[64]; int n; };

__global__ void __launch_bounds__(256)
hd_assemble_kernel(const HdPlan plan, const uint4* __restrict__ tokens, const int* __restrict__ crop_map,
                   const uint4* __restrict__ sep, const uint4* __restrict__ ret, uint4* __restrict__ out, int M, int vecs,
                   long long row0) {
    const long long row = row0 + blockIdx.x;
    int i = 0;
    while (i + 1 < plan.n && row >= plan.img[i + 1].out_row) ++i;      // images are in row order
    const tp_hd_image im = plan.img[i];
    const long long r64 = row - im.out_row;
    const int n = im.h_block * im.w_block, seg_rows = M + 1;
    if (r64 >= (long long)n * seg_rows + (n > 1 ? seg_rows : 0)) return;      // a gap between two images: not ours to write
    const int r = (int)r64;
    const uint4* src;
    if (r < n * seg_rows) {
        const int seg = r / seg_rows, k = r - seg * seg_rows;
        // (crop_map: logical crop -> row block of `tokens`, e.g. the padded slots of a ragged all-gather)
        if (k < M) { const int c = im.first_crop + seg; src = tokens + ((long long)(crop_map ? crop_map[c] : c) * M + k) * vecs; }
        else src = (seg % im.w_block == im.w_block - 1) ? ret : sep;
    } else {
        const int k = r - n * seg_rows, c = im.first_crop + n;
        src = k < M ? tokens + ((long long)(crop_map ? crop_map[c] : c) * M + k) * vecs : ret;
    }
    uint4* dst = out + row * vecs;
    for (int v = threadIdx.x; v < vecs; v += blockDim.x) dst[v] = src[v];
}

int hd_assemble_launch(const tp_hd_image* plan, int n_images, const void* tokens, const int32_t* crop_map, const void* sep,
                       const void* ret, void* out, int M, int D, hipStream_t stream) {
    const int vecs = D / 8;
    for (int base = 0; base < n_images; base += 64) {
        HdPlan hp{};
        hp.n = n_images - base < 64 ? n_images - base : 64;
        for (int i = 0; i < hp.n; ++i) hp.img[i] = plan[base + i];
        const tp_hd_image& last = hp.img[hp.n - 1];
        const long long row0 = hp.img[0].out_row;
        const long long rows = last.out_row + tp_hd_rows(last.h_block, last.w_block, M) - row0;
        if (rows <= 0) continue;
        hipLaunchKernelGGL(hd_assemble_kernel, dim3((unsigned)rows), dim3(256), 0, stream, hp, (const uint4*)tokens,
                           (const int*)crop_map, (const uint4*)sep, (const uint4*)ret, (uint4*)out, M, vecs, row0);
        const int rc = check_launch("hd_assemble_kernel");
        if (rc != TP_OK) return rc;
    }
    return TP_OK;
}

// ---------------------------------------------------------------------------------------------------
// TokenPacker-HD image slicing (reference llava/train/train.py:695-731, duplicated in the eval drivers): the
// normalised image [3, H, W] is resized (bilinear, align_corners = False, no antialias — F.interpolate's
// arithmetic: src = scale (dst + 0.5) - 0.5 clamped at 0, scale = in / out in fp32) to (h_res, w_res), placed
// in the top-left corner of a zero canvas of h_block x w_block blocks of `block` pixels, and the canvas is cut
// into crops (row-major).  With more than one crop a global view follows: the CANVAS (padding included — the
// reference re-uses the variable, train.py:710,728) resized to (hg, wg) and zero-padded to block x block.
// One thread per output pixel; crops [n, 3, block, block] fp32.
__device__ __forceinline__ void bilinear_taps(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256)
hd_slice_crops_kernel(const float* __restrict__ img, int H, int W, int h_block, int w_block, int h_res, int w_res,
                      float* __restrict__ crops, int block) {
    const long long total = (long long)h_block * w_block * 3 * block * block;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % block), py = (int)(idx / block % block), ch = (int)(idx / ((long long)block * block) % 3);
    const int crop = (int)(idx / ((long long)block * block * 3));
    const int Y = (crop / w_block) * block + py, X = (crop % w_block) * block + px;
    float v = 0.f;
    if (Y < h_res && X < w_res) {
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        bilinear_taps((float)H / (float)h_res, Y, H, y0, y1, ly0, ly1);
        bilinear_taps((float)W / (float)w_res, X, W, x0, x1, lx0, lx1);
        const float* c = img + (long long)ch * H * W;
        v = ly0 * (lx0 * c[(long long)y0 * W + x0] + lx1 * c[(long long)y0 * W + x1]) +
            ly1 * (lx0 * c[(long long)y1 * W + x0] + lx1 * c[(long long)y1 * W + x1]);
    }
    crops[idx] = v;
}

__global__ void __launch_bounds__(256)
hd_slice_global_kernel(float* __restrict__ crops, int h_block, int w_block, int hg, int wg, int block) {
    const long long total = (long long)3 * block * block;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % block), py = (int)(idx / block % block), ch = (int)(idx / ((long long)block * block));
    const int CH = h_block * block, CW = w_block * block;
    auto canvas = [&](int Y, int X) -> float {          // the canvas as laid out in the crops just written
        const int crop = (Y / block) * w_block + X / block;
        return crops[(((long long)crop * 3 + ch) * block + Y % block) * block + X % block];
    };
    float v = 0.f;
    if (py < hg && px < wg) {
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        bilinear_taps((float)CH / (float)hg, py, CH, y0, y1, ly0, ly1);
        bilinear_taps((float)CW / (float)wg, px, CW, x0, x1, lx0, lx1);
        v = ly0 * (lx0 * canvas(y0, x0) + lx1 * canvas(y0, x1)) + ly1 * (lx0 * canvas(y1, x0) + lx1 * canvas(y1, x1));
    }
    crops[((long long)h_block * w_block * 3) * block * block + idx] = v;
}

int hd_slice_launch(const float* img, int H, int W, int h_block, int w_block, int h_res, int w_res, int hg, int wg,
                    float* crops, int block, hipStream_t stream) {
    const long long total = (long long)h_block * w_block * 3 * block * block;
    hipLaunchKernelGGL(hd_slice_crops_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, img, H, W,
                       h_block, w_block, h_res, w_res, crops, block);
    int rc = check_launch("hd_slice_crops_kernel");
    if (rc != TP_OK || h_block * w_block <= 1) return rc;
    const long long gtotal = (long long)3 * block * block;
    hipLaunchKernelGGL(hd_slice_global_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, stream, crops,
                       h_block, w_block, hg, wg, block);
    return check_launch("hd_slice_global_kernel");
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm statistics: the producing GEMM leaves one (mean, M2) slab per 128 output columns ([parts][M][2]; M2 = sum
// of squared deviations from the slab's own mean); this merges them (Chan's formula, slab order -> deterministic)
// into per-row (mean, rstd) for the consuming GEMM's epilogue.  Unlike E[x^2] - mean^2 this keeps its accuracy when
// |mean| >> std (nn.LayerNorm itself uses a two-pass / Welford reduction).  ~10 MB of traffic at B=256: noise.
// NPARTS > 0: the slab count is a compile-time constant — all slabs of a row are fetched at once (independent loads in
// flight, one pass) instead of one dependent HBM round trip per slab (measured 58 us per call at B = 256 that way).
template <int NPARTS>
__global__ void __launch_bounds__(256)
ln_finalize_kernel(const float* __restrict__ parts, float* __restrict__ mean_rstd, long long M, int nparts,
                   float inv_dim, float eps, bool second_moment, bool pair_rstd) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (m >= M) return;
    const float* pg = parts + (long long)g * nparts * M * 2;
    if constexpr (NPARTS > 0) {
        float2 r = ln_merge_slabs<NPARTS>(pg, M, m, inv_dim, eps, second_moment);
        // pair_rstd (second-moment statistics only: every mean is 0): the mean slot of group 1 carries group 0's rstd — what the
        // V launch of the decoupled attention epilogues scales the K launch's raw logits with (GemmArgs::attn_decoupled)
        if (pair_rstd && g == 1) r.x = ln_merge_slabs<NPARTS>(parts, M, m, inv_dim, eps, second_moment).y;
        *(float2*)(mean_rstd + ((long long)g * M + m) * 2) = r;
        return;
    } else {
        float s1 = 0.f, q = 0.f, between = 0.f, mu;
        for (int pp = 0; pp < nparts; ++pp) {
            const float2 st = *(const float2*)(pg + ((long long)pp * M + m) * 2);
            s1 += st.x; q += st.y;
        }
        mu = s1 / (float)nparts;
        for (int pp = 0; pp < nparts; ++pp) {
            const float d = pg[((long long)pp * M + m) * 2] - mu;
            between = fmaf(d, d, between);
        }
        const float var = (q + 128.0f * between) * inv_dim;    // biased variance (nn.LayerNorm); >= 0 by construction
        *(float2*)(mean_rstd + ((long long)g * M + m) * 2) = second_moment ? make_float2(0.f, 1.0f / sqrtf(fmaf(mu, mu, var) + eps))
                                                                          : make_float2(mu, 1.0f / sqrtf(var + eps));
    }
}

int ln_finalize_launch(const float* parts, float* mean_rstd, long long M, int nparts, int groups, int ln_dim,
                       float eps, hipStream_t stream, bool second_moment, bool pair_rstd) {
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)groups);
    if (pair_rstd && (nparts != 8 || groups != 2 || !second_moment)) { set_error("ln_finalize: pair_rstd needs 8 slabs, 2 groups, second-moment statistics"); return TP_ERR_INVALID_ARG; }
    if (nparts == 8)
        hipLaunchKernelGGL(ln_finalize_kernel<8>, grid, dim3(256), 0, stream, parts, mean_rstd, M, nparts, 1.0f / ln_dim, eps, second_moment, pair_rstd);
    else
        hipLaunchKernelGGL(ln_finalize_kernel<0>, grid, dim3(256), 0, stream, parts, mean_rstd, M, nparts, 1.0f / ln_dim, eps, second_moment, false);
    return check_launch("ln_finalize_kernel");
}

// ---------------------------------------------------------------------------------------------------
// Second half of a K-split GEMM (TP_TUNE_SPLIT_K, small batches): the S fp32 partial products [S][M][N] of the 128-tile
// kernel's grouped launch are summed in split order (deterministic; NOT the summation order of the unsplit kernels), then
// bias, the erf GELU of tp_gemm_common.h, fp16 saturation and the cast — what the fused epilogue would have done.
} // namespace tp
#include "tp_gemm_common.h"
namespace tp {
template <typename TO>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ part, int S, long long MN, int N, const float* __restrict__ bias, int gelu,
                     TO* __restrict__ out, long long ldc, int split_cols, long long split_stride, int* __restrict__ sat_flag,
                     int sat_bit) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= MN) return;
    const long long m = i4 / N;
    const int n = (int)(i4 - m * N);
    f32x4 acc = *(const f32x4*)(part + i4);
    for (int sp = 1; sp < S; ++sp) acc += *(const f32x4*)(part + (long long)sp * MN + i4);
    if (bias) acc += *(const f32x4*)(bias + n);
    if (gelu) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2_ev gv = gelu_erf2(f32x2_ev{acc[r], acc[r + 1]});
            acc[r] = gv[0]; acc[r + 1] = gv[1];
        }
    }
    TO* o = out + m * ldc + n;
    if (split_cols > 0 && n >= split_cols) o += split_stride - split_cols;      // second output slab (GemmArgs::c_split_cols)
    if constexpr (std::is_same<TO, float>::value) {
        *(f32x4*)o = acc;
    } else {
        if constexpr (std::is_same<TO, f16_t>::value) {
            // the sticky fp16-saturation report of the GEMM epilogues (GemmArgs::sat_flag), for the K-split route as well: a value
            // the reference's fp16 arithmetic would have turned into inf — or a NaN (`!(|v| < 65520)` is true for it) — ORs the
            // stage's bit.  (This kernel is memory-bound and runs for batches of <= 8 images: the test per element is free.)
            bool sat = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sat |= !(fabsf(acc[r]) < 65520.f);
                acc[r] = __builtin_amdgcn_fmed3f(acc[r], -65504.f, 65504.f);
            }
            if (sat_flag && sat) __hip_atomic_fetch_or(sat_flag, sat_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        using O4 = typename Vec<TO>::x4;
        *(O4*)o = __builtin_convertvector(acc, O4);
    }
}

int splitk_reduce_launch(const float* partials, int S, int M, int N, const float* bias, int gelu, void* out, long long ldc,
                         int out_dtype, hipStream_t stream, int split_cols, long long split_stride, int* sat_flag, int sat_bit) {
    const long long MN = (long long)M * N;
    const unsigned blocks = (unsigned)((MN / 4 + 255) / 256);
    if (out_dtype == TP_F16)
        hipLaunchKernelGGL(splitk_reduce_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, partials, S, MN, N, bias, gelu, (f16_t*)out, ldc, split_cols, split_stride, sat_flag, sat_bit);
    else if (out_dtype == TP_BF16)
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, partials, S, MN, N, bias, gelu, (bf16_t*)out, ldc, split_cols, split_stride, nullptr, 0);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, partials, S, MN, N, bias, gelu, (float*)out, ldc, split_cols, split_stride, nullptr, 0);
    return check_launch("splitk_reduce_kernel");
}

// ---------------------------------------------------------------------------------------------------
// One-time weight preparation.
template <typename T>
__global__ void pack_cast_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

int pack_cast_f32_launch(int dtype, const void* src, float* dst, int n, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(pack_cast_f32_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, dst, n);
    else
        hipLaunchKernelGGL(pack_cast_f32_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, (const f16_t*)src, dst, n);
    return check_launch("pack_cast_f32_kernel");
}

// (a clamped element is an event that never happens with trained weights: one atomic per offender is fine)
__device__ __forceinline__ float clamp_f16_range(float v, int* sat) {
    if (sat && !(fabsf(v) <= 65504.f)) atomicAdd(sat, 1);
    return fminf(fmaxf(v, -65504.f), 65504.f);
}

template <typename T>
__global__ void pack_cast_f16_kernel(const T* __restrict__ src, f16_t* __restrict__ dst, long long n, int* sat) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (f16_t)clamp_f16_range((float)src[i], sat);   // exact for in-range bf16
}

int pack_cast_f16_launch(int dtype, const void* src, void* dst, long long n, hipStream_t stream, int* sat) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(pack_cast_f16_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, (f16_t*)dst, n, sat);
    else
        hipLaunchKernelGGL(pack_cast_f16_kernel<f16_t>, dim3(blocks), dim3(256), 0, stream, (const f16_t*)src, (f16_t*)dst, n, sat);
    return check_launch("pack_cast_f16_kernel");
}

// A batch of small element-wise operations in ONE launch (round 6): the per-step weight pack of a training step is ~20 casts and
// copies of 1 K .. 17 M elements — 4-27 us each, most of it launch latency, 0.15 ms per step together (5 % of a 32-image training
// step, profiles/r06n_train_b32_kernel_stats.csv).  blockIdx.y picks the operation, the x blocks stride over its elements in
// 8-element vectors (every operand of the pack is 16-byte aligned with a multiple of 8 elements; anything else takes the scalar path).
template <typename T>
__global__ void __launch_bounds__(256)
pack_batch_kernel(const BatchOps ops, int* __restrict__ sat) {
    const BatchOp o = ops.op[blockIdx.y];
    using S8 = typename Vec<T>::x8;
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (o.n & 7) == 0 && (((uintptr_t)o.src | (uintptr_t)o.dst) & 15) == 0;
    if (vec) {
        for (long long i = t0; i < o.n / 8; i += stride) {
            const S8 v = *((const S8*)o.src + i);
            if (o.kind == BATCH_OP_COPY16) { *((S8*)o.dst + i) = v; }
            else if (o.kind == BATCH_OP_TO_F32) {
                f32x4 a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = (float)v[e]; b[e] = (float)v[4 + e]; }
                *((f32x4*)o.dst + 2 * i) = a; *((f32x4*)o.dst + 2 * i + 1) = b;
            } else {
                f16x8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (f16_t)clamp_f16_range((float)v[e], sat);
                *((f16x8*)o.dst + i) = r;
            }
        }
    } else {
        for (long long i = t0; i < o.n; i += stride) {
            const T v = ((const T*)o.src)[i];
            if (o.kind == BATCH_OP_COPY16) ((T*)o.dst)[i] = v;
            else if (o.kind == BATCH_OP_TO_F32) ((float*)o.dst)[i] = (float)v;
            else ((f16_t*)o.dst)[i] = (f16_t)clamp_f16_range((float)v, sat);
        }
    }
}

int pack_batch_launch(int dtype, const BatchOps& ops, hipStream_t stream, int* sat) {
    if (ops.count <= 0) return TP_OK;
    if (ops.count > kBatchOps) { set_error("pack batch: %d operations > %d", ops.count, kBatchOps); return TP_ERR_INVALID_ARG; }
    long long nmax = 0;
    for (int i = 0; i < ops.count; ++i) nmax = ops.op[i].n > nmax ? ops.op[i].n : nmax;
    long long bx = (nmax / 8 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3((unsigned)bx, (unsigned)ops.count), dim3(256), 0, stream, ops, sat);
    else
        hipLaunchKernelGGL(pack_batch_kernel<f16_t>, dim3((unsigned)bx, (unsigned)ops.count), dim3(256), 0, stream, ops, sat);
    return check_launch("pack_batch_kernel");
}

// Debug scan of an fp16 activation buffer for saturated epilogue outputs (tp_debug_count_saturated): every kernel of
// the path clamps to +-65504 instead of producing inf, so an element AT the bound (or a NaN) marks a clamp.
__global__ void __launch_bounds__(256)
count_saturated_kernel(const f16x8* __restrict__ buf, long long nvec, int* __restrict__ count) {
    int local = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        const f16x8 v = buf[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) local += !(fabsf((float)v[e]) < 65504.f);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

int count_saturated_launch(const void* buf, long long n, int* count, hipStream_t stream) {
    const long long nvec = n / 8;
    long long blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(count_saturated_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const f16x8*)buf, nvec, count);
    return check_launch("count_saturated_kernel");
}

// out_proj folded into mlp[0] (both are linear with nothing in between, builder.py:126-130 -> :136):
//   A2 = GELU((O·Wout^T + bout)·Wm0^T + bm0) = GELU(O·(Wm0·Wout)^T + (Wm0·bout + bm0)).
// The [D,1024]x[1024,1024] product runs on the path's own MFMA kernel (fp32 out) against a transposed copy of
// Wout; these three helpers are the transposition, the saturating fp32->fp16 rounding and the bias fold.
__global__ void __launch_bounds__(256)
pack_transpose_f16_kernel(const f16_t* __restrict__ src, f16_t* __restrict__ dst, int n) {
    __shared__ f16_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(long long)(by + r) * n + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) dst[(long long)(bx + r) * n + by + tx] = tile[tx][r];
}

int pack_transpose_f16_launch(const void* src, void* dst, int n, hipStream_t stream) {
    hipLaunchKernelGGL(pack_transpose_f16_kernel, dim3(n / 32, n / 32), dim3(256), 0, stream, (const f16_t*)src,
                       (f16_t*)dst, n);
    return check_launch("pack_transpose_f16_kernel");
}

__global__ void pack_round_f16_kernel(const float* __restrict__ src, f16_t* __restrict__ dst, long long n, int* sat) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (f16_t)clamp_f16_range(src[i], sat);
}

int pack_round_f16_launch(const float* src, void* dst, long long n, hipStream_t stream, int* sat) {
    hipLaunchKernelGGL(pack_round_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, (f16_t*)dst, n, sat);
    return check_launch("pack_round_f16_kernel");
}

__global__ void __launch_bounds__(256)
pack_bias_fold_kernel(const f16_t* __restrict__ w, const float* __restrict__ v, const float* __restrict__ b,
                      float* __restrict__ out, int n_in) {
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int kk = threadIdx.x; kk < n_in; kk += blockDim.x) acc = fmaf((float)w[(long long)n * n_in + kk], v[kk], acc);
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = red[0] + red[1] + red[2] + red[3] + (b ? b[n] : 0.f);
}

int pack_bias_fold_launch(const void* w, const float* v, const float* b, float* out, int n_out, int n_in,
                          hipStream_t stream) {
    hipLaunchKernelGGL(pack_bias_fold_kernel, dim3(n_out), dim3(256), 0, stream, (const f16_t*)w, v, b, out, n_in);
    return check_launch("pack_bias_fold_kernel");
}

// LayerNorm folded into the linear that follows it.  For y = LN(h)·W^T + b with
// LN(h) = (h − mu)·rstd·gamma + beta:
//     y_n = rstd·( Σ_k h_k W'_nk − mu·c_n ) + b'_n,   W'_nk = W_nk·gamma_k (rounded to fp16),
//     c_n = Σ_k W'_nk (of the ROUNDED W', so the mean term cancels exactly),  b'_n = Σ_k beta_k W_nk + b_n.
// One workgroup per output row n.
template <typename T>
__global__ void __launch_bounds__(256)
pack_ln_fold_kernel(const T* __restrict__ w, const T* __restrict__ bias, const T* __restrict__ gamma,
                    const T* __restrict__ beta, f16_t* __restrict__ w_out, float* __restrict__ colsum,
                    float* __restrict__ bias_out, int n_in, int* sat) {
    const int n = blockIdx.x;
    float cs = 0.f, bs = 0.f;
    for (int kk = threadIdx.x; kk < n_in; kk += blockDim.x) {
        const float wv = (float)w[(long long)n * n_in + kk];
        const f16_t wp = (f16_t)clamp_f16_range(wv * (float)gamma[kk], sat);
        w_out[(long long)n * n_in + kk] = wp;
        cs += (float)wp;
        bs = fmaf((float)beta[kk], wv, bs);
    }
    __shared__ float red[2][4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { cs += __shfl_xor(cs, off); bs += __shfl_xor(bs, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cs; red[1][threadIdx.x >> 6] = bs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        colsum[n] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        bias_out[n] = red[1][0] + red[1][1] + red[1][2] + red[1][3] + (float)bias[n];
    }
}

// The part of the folded weight its fp16 rounding drops.  W'_nk = W_nk·gamma_k is EXACT in fp32 (a product of two 16-bit floats);
// pack_ln_fold stores hi = fp16(W'); this kernel writes lo = fp16(W' − hi) (the residual to ~2^-22), c_exact[n] = Σ_k W'_nk and,
// with v, d_exact[n] = Σ_k W'_nk v_k — what the pre-multiplied chain weights of the centred LayerNorm chain are built from
// (tp_pack_weights: Wcc = (hi + lo)·W2 − c_exact ⊗ w̄), so that the fold's own rounding no longer sits in series with theirs.
template <typename T>
__global__ void __launch_bounds__(256)
pack_ln_fold_residual_kernel(const T* __restrict__ w, const T* __restrict__ gamma, const float* __restrict__ v,
                             f16_t* __restrict__ lo, float* __restrict__ c_exact, float* __restrict__ d_exact, int n_in) {
    const int n = blockIdx.x;
    float cs = 0.f, ds = 0.f;
    for (int kk = threadIdx.x; kk < n_in; kk += blockDim.x) {
        const float we = (float)w[(long long)n * n_in + kk] * (float)gamma[kk];
        const float hi = (float)(f16_t)fminf(fmaxf(we, -65504.f), 65504.f);
        lo[(long long)n * n_in + kk] = (f16_t)(we - hi);
        cs += we;
        if (v) ds = fmaf(we, v[kk], ds);
    }
    __shared__ float red[2][4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { cs += __shfl_xor(cs, off); ds += __shfl_xor(ds, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cs; red[1][threadIdx.x >> 6] = ds; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c_exact[n] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        if (d_exact) d_exact[n] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

int pack_ln_fold_residual_launch(int dtype, const void* w, const void* gamma, const float* v, void* lo_f16, float* c_exact,
                                 float* d_exact, int n_out, int n_in, hipStream_t stream) {
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(pack_ln_fold_residual_kernel<bf16_t>, dim3(n_out), dim3(256), 0, stream, (const bf16_t*)w,
                           (const bf16_t*)gamma, v, (f16_t*)lo_f16, c_exact, d_exact, n_in);
    else
        hipLaunchKernelGGL(pack_ln_fold_residual_kernel<f16_t>, dim3(n_out), dim3(256), 0, stream, (const f16_t*)w,
                           (const f16_t*)gamma, v, (f16_t*)lo_f16, c_exact, d_exact, n_in);
    return check_launch("pack_ln_fold_residual_kernel");
}

int pack_ln_fold_launch(int dtype, const void* w, const void* bias, const void* gamma,
                        const void* beta, void* w_out, float* colsum, float* bias_out, int n_out,
                        int n_in, hipStream_t stream, int* sat) {
    if (dtype == TP_BF16)
        hipLaunchKernelGGL(pack_ln_fold_kernel<bf16_t>, dim3(n_out), dim3(256), 0, stream,
                           (const bf16_t*)w, (const bf16_t*)bias, (const bf16_t*)gamma, (const bf16_t*)beta,
                           (f16_t*)w_out, colsum, bias_out, n_in, sat);
    else
        hipLaunchKernelGGL(pack_ln_fold_kernel<f16_t>, dim3(n_out), dim3(256), 0, stream,
                           (const f16_t*)w, (const f16_t*)bias, (const f16_t*)gamma, (const f16_t*)beta,
                           (f16_t*)w_out, colsum, bias_out, n_in, sat);
    return check_launch("pack_ln_fold_kernel");
}

}  // namespace tp
