// tp_gemm.hip — the dense contractions of the TokenPacker path (99.98 % of its FLOPs)
// as one hand-written MFMA kernel family for gfx950 (MI355X, CDNA4).
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]^T )          A, W: bf16/fp16, K-contiguous (nn.Linear layout)
//
// Replaces the 11 nn.Linear / F.linear calls of reference
// llava/model/multimodal_projector/builder.py:112,113,120,126-130,136 together with the GELU,
// LayerNorm and dtype-cast kernels between them (fused into the epilogue).
//
// Structure (CDNA4-first, not a CUDA warp tiling):
//   * 64-wide wavefronts, v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate.
//   * block tile BMxBNx64; A and W K-slabs are DMA'd HBM->LDS with global_load_lds_dwordx4
//     (1 KiB per wave instruction = 8 rows x 128 B), double buffered, one barrier per K-slab.
//   * LDS image is lane-linear (a DMA constraint), so the bank-conflict swizzle
//     slot' = slot ^ (row & 7) is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 fragment address (same involution on both sides).
//   * operands are swapped into the MFMA (W rows as the "A" fragment) so each lane ends up with
//     4 CONSECUTIVE output columns of one row -> 8-byte packed stores, float4 bias loads.
//   * XCD-aware tile order: the 8 XCDs own contiguous ranges of the tile list so that the tiles
//     sharing an A row-panel hit the same private L2.
//   * epilogue: LayerNorm-fold (a LayerNorm in front of the linear applied as
//     rstd·(acc − mu·colsum) from per-row (mean, rstd)), bias, erf-form GELU,
//     per-row (sum, sumsq) partials of the rounded output for the NEXT LayerNorm, cast.
#include "tp_gemm_common.h"
#include <mutex>

namespace tp {

template <int BM, int BN, int NST = 2>
constexpr int gemm_lds_bytes() { return NST * (BM + BN) * ROW_BYTES; }

// STRIDED_A: A rows live in batches of rows_per_batch rows with a batch stride (the CLIP tower's
// [:,1:] slices) — only the first K/V layer needs it, which also gives that launch (45 % of the path's
// FLOPs) its own kernel symbol in profiles.
// TI: operand element type (bf16 / fp16) — selects the MFMA;  TO: output element type (bf16 / fp16 / float).
// NST: K-slab buffers in the LDS ring; NST - 1 slabs are in flight while one is consumed.  2 = the double buffer every full-chip
// launch uses (two workgroups per CU hide each other's slab round trip); 4 = the small-batch form (one workgroup per CU, 8 waves:
// a K step is ~0.1 us of MFMA behind a ~0.5 us slab round trip — with one slab in flight the K loop IS that round trip).
template <typename TI, typename TO, int BM, int BN, int WM, int WN, int AMODE, bool TRAIN_EPI, int XMODE = 0, int NST = 2>   // XMODE: tp_gemm8.hip
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
gemm_kernel(const GemmArgs p, const int tiles_n, const int xcd_swizzle) {
    static_assert(NST == 2 || NST == 4, "ring depth");
    using T = TI;
    using X8 = typename Vec<TI>::x8;
    constexpr int NWM = BM / WM, NWN = BN / WN, NW = NWM * NWN;
    constexpr int FM = WM / 16, FN = WN / 16;          // 16x16 fragments per wave
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
    constexpr int CHUNKS = (BM + BN) / 8;              // 1 KiB DMA pieces per K-slab
    constexpr int CPW = CHUNKS / NW;                   // pieces per wave
    static_assert(CHUNKS % NW == 0, "DMA pieces must divide over the waves");
    static_assert((BM / 8) % CPW == 0, "a wave's pieces must not straddle the A/W boundary");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    // ---- tile assignment -----------------------------------------------------------------------
    int bid = blockIdx.x;
    if (xcd_swizzle) bid = xcd_remap(bid, gridDim.x);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int g = blockIdx.y;

    const char* __restrict__ Ag = p.A + g * p.a_gs;
    const char* __restrict__ Wg = p.W + g * p.w_gs;

    // ---- per-lane DMA source pointers ------------------------------------------------------------
    // piece c covers tile rows 8c..8c+7 (A rows first, then W rows); lane -> row 8c + lane/8,
    // LDS slot' = lane%8, logical K slot = slot' ^ (row & 7) = (lane%8) ^ (lane/8).
    const int kslot = (lane & 7) ^ (lane >> 3);
    const char* src[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = wave * CPW + i;
        if (c < BM / 8) {
            int row = m0 + c * 8 + (lane >> 3);
            row = row < p.M ? row : p.M - 1;
            if constexpr (AMODE != 0) {
                if (p.a_region_s > 0) row = region_major_to_raster(row, p.a_region_g, p.a_region_s);
                const int b = row / p.rows_per_batch;
                const int t = row - b * p.rows_per_batch;
                src[i] = Ag + (long long)b * p.a_batch_stride_bytes + (long long)t * p.lda_bytes + kslot * 16;
            } else {
                src[i] = Ag + (long long)row * p.lda_bytes + kslot * 16;
            }
        } else {
            const int n = n0 + (c - BM / 8) * 8 + (lane >> 3);
            src[i] = Wg + (long long)n * (p.ldw_bytes ? p.ldw_bytes : (long long)p.K * 2) + kslot * 16;
        }
    }

    // statistics-only launches on an UPPER-TRIANGULAR weight (GemmArgs::tri): W[n][k] = 0 for k < n0 — start at K-tile n0 / 64
    const int kt0 = (XMODE == 1 && p.tri) ? n0 / BK : 0;
    auto issue = [&](int kt_rel, int stage) {
        const int kt = kt_rel + kt0;
        char* dst = smem + stage * STAGE + wave * CPW * 1024;
        long long a_adv = (long long)kt * ROW_BYTES;            // K advance of the A pieces
        if constexpr (AMODE == 0) {                              // GemmArgs::a_k_dup: the leading K-tiles of A serve two K-tiles of W each
            const int w = p.a_k_dup / BK;
            a_adv = (long long)(kt < 2 * w ? kt >> 1 : kt - w) * ROW_BYTES;
        }
        if constexpr (AMODE == 2) {                              // K split over four source tensors
            // (a K-split launch — groups over K, a_gs = 0 — walks the sources with the group's GLOBAL K-tile index)
            const int ktg = kt + (p.parts_k_groups ? g * (p.K / BK) : 0);
            const int tpp = p.k_part / BK, part = ktg / tpp;
            a_adv = (p.A_parts[part] - p.A_parts[0]) + (long long)(ktg - part * tpp) * ROW_BYTES;
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const bool is_a = (wave * CPW + i) < BM / 8;         // wave-uniform
            __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + (is_a ? a_adv : (long long)kt * ROW_BYTES)),
                                             (lds_void*)(dst + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (swizzled) ---------------------------------------------------------
    // fragment rows are 16-aligned, so row & 7 == lane & 7.
    const int frag_off0 = (lane & 15) * ROW_BYTES + ((((lane >> 4)) ^ (lane & 7)) << 4);
    const int frag_off1 = (lane & 15) * ROW_BYTES + (((4 + (lane >> 4)) ^ (lane & 7)) << 4);

    // the first K-slabs go out before anything else touches memory: the parameter loads below then wait WITH them
    const int nk = p.K / BK - kt0;
    issue(0, 0);
    if constexpr (NST > 2) {
#pragma unroll
        for (int s_ = 1; s_ < NST - 1; ++s_) if (s_ < nk) issue(s_, s_);
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (XMODE >= 2) {                         // accumulators start from a per-column constant (GemmArgs::acc_init)
        const float* __restrict__ ai = p.acc_init + g * p.acc_init_gs + n0 + wn * WN + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 dv = *(const f32x4*)(ai + j * 16);
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[i][j] = dv;
        }
    }

    // LayerNorm-fold operands (per-row mean, rstd) do not depend on the contraction: fetch them now so
    // their latency hides under the K loop instead of serialising the epilogue.
    float2 mean_rstd[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) mean_rstd[i] = make_float2(0.f, 1.f);
    if (p.flags & TP_LINEAR_LN_FOLD) {
        if (p.stats_parts) {                            // the producer's slabs, merged here (no ln_finalize launch):
            float2 st[FM][8];                           // every load of every row first, ONE wait, then the arithmetic
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                int m = m0 + wm * WM + i * 16 + (lane & 15);
                m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp)
                    st[i][pp] = *(const float2*)(p.stats_parts + g * p.stats_parts_gs + ((long long)pp * p.M + m) * 2);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) mean_rstd[i] = ln_merge_values<8>(st[i], p.ln_inv_dim, p.ln_eps, p.ln_second_moment != 0);
            if constexpr (XMODE == 4) {
                if (p.attn_kstats_parts) {              // decoupled attention: the K rows' rstd rides in the mean slot (GemmArgs::attn_decoupled)
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        int m = m0 + wm * WM + i * 16 + (lane & 15);
                        m = m < p.M ? m : p.M - 1;
#pragma unroll
                        for (int pp = 0; pp < 8; ++pp) st[i][pp] = *(const float2*)(p.attn_kstats_parts + ((long long)pp * p.M + m) * 2);
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i) mean_rstd[i].x = ln_merge_values<8>(st[i], p.ln_inv_dim, p.ln_eps, p.ln_second_moment != 0).y;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                int m = m0 + wm * WM + i * 16 + (lane & 15);
                m = m < p.M ? m : p.M - 1;
                mean_rstd[i] = *(const float2*)(p.stats_in + g * p.stats_in_gs + (long long)m * 2);
            }
        }
    }

    for (int kt = 0; kt < nk; ++kt) {
        // slab kt has landed for every wave; every wave is done reading the buffer the next issue overwrites
        if constexpr (NST == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            // slabs kt + 1 .. kt + NST - 2 stay in flight (CPW DMA instructions per wave each; everything older — slab kt and
            // the epilogue parameters fetched behind the prologue — has retired by then: vmcnt retires in order)
            const int ahead = nk - 1 - kt;
            if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * CPW) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // (a __syncthreads() would drain vmcnt)
        }
        if (kt + NST - 1 < nk) issue(kt + NST - 1, (kt + NST - 1) & (NST - 1));

        const char* sA = smem + (kt & (NST - 1)) * STAGE + wm * WM * ROW_BYTES;
        const char* sB = smem + (kt & (NST - 1)) * STAGE + A_BYTES + wn * WN * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = ks ? frag_off1 : frag_off0;
            X8 a[FM], b[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = *(const X8*)(sB + j * 16 * ROW_BYTES + off);
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = *(const X8*)(sA + i * 16 * ROW_BYTES + off);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = Mma<T>::run(b[j], a[i], acc[i][j]);   // swapped: lane gets 4 consecutive n
        }
    }

    if constexpr (XMODE == 3)
        attn_logits_epilogue<BM, BN, WM, WN, false>(acc, p, g, m0, n0, wm, wn, lane, tid, mean_rstd, smem);
    else if constexpr (XMODE == 4)
        attn_sum_epilogue<BM, BN, WM, WN, false>(acc, p, g, m0, n0, wm, wn, lane, mean_rstd);
    else
        gemm_epilogue<TO, BM, BN, WM, WN, false, TRAIN_EPI, XMODE == 1>(acc, p, g, m0, n0, tile_n, wm, wn, lane, tid, mean_rstd, smem);
}

// ---- host side ------------------------------------------------------------------------------------
int gemm_pick_tile(int M, int N, int forced, int groups) {
    if (forced == 0) forced = tuning(TP_TUNE_GEMM_TILE);
    if (forced == 2 || forced == 3 || forced == 4) forced = 0;      // (half / 192-row tiles and their A/B switch: gemm_route's business)
    if (forced == 128) return 128;
    if (forced == 256 && N % 256 == 0) return 256;
    if (N % 256 != 0) return 128;
    const long long tiles256 = (long long)((M + 255) / 256) * (N / 256) * (groups > 0 ? groups : 1);
    return tiles256 >= 200 ? 256 : 128;     // the persistent 256-tile kernel from ~0.8 of a CU round up, else finer tiles
}

// The 128 x 128 tile exists as 4 waves of 64 x 64 (two workgroups = 8 waves per CU) and as 8 waves of 32 x 64 (16 waves per CU):
// a launch that leaves most CUs with ONE workgroup (a small batch) runs one wave per SIMD with the first and hides neither
// its LDS nor its barrier latency.  Same MFMA shape, same K order: bit-identical.  WIDE selects the 8-wave form.
template <typename TI, typename TO, int BM, int BN, int WM, int WN, int AMODE, bool TRAIN_EPI = false, int XMODE = 0, bool WIDE = false, int NST = 2>
static int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    if constexpr (!WIDE && BM == 128 && BN == 128 && WM == 64 && WN == 64) {
        const long long wgs = (long long)((a.M + BM - 1) / BM) * (a.N / BN) * (a.groups > 0 ? a.groups : 1);
        const int mode = tuning(TP_TUNE_SMALL_GEMM_WAVES);
        if (mode == 8 || mode == 9 || (mode == 0 && wgs <= 512)) {
            // at most one workgroup per CU: the four-slab ring (mode 9 forces the double buffer for an A/B, bit-identical)
            if constexpr (!TRAIN_EPI)
                if (mode != 9 && wgs <= gemm8_persistent_cus())
                    return launch_cfg<TI, TO, BM, BN, 32, 64, AMODE, TRAIN_EPI, XMODE, true, 4>(a, stream);
            return launch_cfg<TI, TO, BM, BN, 32, 64, AMODE, TRAIN_EPI, XMODE, true>(a, stream);
        }
    }
    constexpr int lds = gemm_lds_bytes<BM, BN, NST>();
    constexpr int threads = (BM / WM) * (BN / WN) * 64;
    auto kern = gemm_kernel<TI, TO, BM, BN, WM, WN, AMODE, TRAIN_EPI, XMODE, NST>;
    static DynLdsAttr attr;                             // (per device, a failure is not cached: tp_internal.h)
    const hipError_t attr_err = attr.ensure(reinterpret_cast<const void*>(kern), lds);
    if (attr_err != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", lds, hipGetErrorString(attr_err));
        return TP_ERR_LAUNCH;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stream, a, tiles_n, tuning(TP_TUNE_XCD_SWIZZLE));
    return check_launch("gemm_kernel");
}

template <typename TI, typename TO>
static int launch_types(const GemmArgs& a, hipStream_t stream) {
    const int tile = gemm_pick_tile(a.M, a.N, a.tile, a.groups);
    const bool strided = a.rows_per_batch < a.M || a.a_region_s > 0;     // (region-major rows: the strided-A kernels)
    constexpr bool HALF_OUT = !std::is_same<TO, float>::value;
    const bool train_epi = (a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD)) != 0;
    if ((a.flags & TP_LINEAR_NO_STORE) || a.acc_init) {     // the two GEMMs of the fused LayerNorm chain (128-tile form)
        if constexpr (std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value) {
            if (strided || train_epi || ((a.flags & TP_LINEAR_NO_STORE) && a.acc_init) || a.A_parts[0]) {
                set_error("tp gemm: NO_STORE / acc_init take a contiguous A, no training epilogue, and not both at once");
                return TP_ERR_INVALID_ARG;
            }
            // (the absorbed schedule's per-head V GEMM — u fp16, W as K-tile pairs hi_t | lo_t, K = 2 E — arrives here as a contiguous A with GemmArgs::a_k_dup)
            if (a.attn_mode)                                 // (validated by gemm_launch)
                return a.attn_mode == 1 ? launch_cfg<TI, TO, 128, 128, 64, 64, 0, false, 3>(a, stream)
                                        : launch_cfg<TI, TO, 128, 128, 64, 64, 0, false, 4>(a, stream);
            return (a.flags & TP_LINEAR_NO_STORE) ? launch_cfg<TI, TO, 128, 128, 64, 64, 0, false, 1>(a, stream)
                                                  : launch_cfg<TI, TO, 128, 128, 64, 64, 0, false, 2>(a, stream);
        }
        set_error("tp gemm: NO_STORE / acc_init are built for fp16 operands and output");
        return TP_ERR_INVALID_ARG;
    }
    if (a.A_parts[0] || train_epi) {                   // training epilogues / four-source A: 128-tile kernel only
        if constexpr (HALF_OUT) {
            if (a.A_parts[0]) {
                if constexpr (std::is_same<TO, f16_t>::value)
                    return train_epi ? launch_cfg<TI, TO, 128, 128, 64, 64, 2, true>(a, stream)
                                     : launch_cfg<TI, TO, 128, 128, 64, 64, 2, false>(a, stream);
                set_error("tp gemm: a multi-part A operand is supported for fp16 (or K-split fp32) output only");
                return TP_ERR_INVALID_ARG;
            }
            return strided ? launch_cfg<TI, TO, 128, 128, 64, 64, 1, true>(a, stream)
                           : launch_cfg<TI, TO, 128, 128, 64, 64, 0, true>(a, stream);
        } else {
            if (a.A_parts[0] && !train_epi && a.parts_k_groups)      // the fp32 partials of a K-split first layer (tp_api.hip)
                return launch_cfg<TI, TO, 128, 128, 64, 64, 2, false>(a, stream);
            set_error("tp gemm: training epilogues / multi-part A need a 16-bit output");
            return TP_ERR_INVALID_ARG;
        }
    }
    if (tile == 256) { set_error("tp gemm: 256-column tiles are served by the ping-pong kernel (tp_gemm8.hip)"); return TP_ERR_INVALID_ARG; }
    return strided ? launch_cfg<TI, TO, 128, 128, 64, 64, 1>(a, stream)
                   : launch_cfg<TI, TO, 128, 128, 64, 64, 0>(a, stream);
}

// Where gemm_launch sends a K-contiguous problem.
// Tile shape by a round count.  256x256 tiles are the scheduling grain of the persistent kernel: a last round that
// fills a fraction of the chip costs a whole tile time.  When the choice is ours (no forced tile / kernel), compare
//   (a) all full tiles, (b) the full rounds as full tiles + the remaining rows as 128x256 half tiles (a second
//   launch over a row window), (c) all half tiles
// with a half tile at 0.75 of a full tile's time (0.63 at K <= 1024; a one-round tail also pays its first tile's
// un-hidden DMA latency) and a second launch at ~8 us (dependent-launch gap) relative to a full tile's
// 8.4 + 1.47 K/64 us (profiles/README.md).  Same epilogue, bit-identical results whatever the shape.
enum { ROUTE_SMALL = 0, ROUTE_G8 = 1, ROUTE_G8_HALF = 2, ROUTE_G8_SPLIT = 3, ROUTE_G8_192 = 4 };
static int gemm_route(const GemmArgs& a, long long* head_rows_out) {
    const int tile_knob = tuning(TP_TUNE_GEMM_TILE);       // 0 auto | 2 all half tiles | 3 all 192-row tiles | 4 auto without 192-row tiles (A/B)
    const bool free_choice = a.tile == 0 && (tile_knob == 0 || tile_knob == 4) &&
                             a.m_begin == 0 && a.m_end == 0 && !a.half_tiles && !a.A_parts[0] && a.N % 256 == 0 && a.K >= 2 * BK;
    if (free_choice) {
        const int cus = gemm8_persistent_cus(), groups = a.groups > 0 ? a.groups : 1;
        // (one group only: the workgroups of a second group start on the CUs the first group's last round leaves idle,
        // so a grouped launch has no empty tail to fill — measured: no gain on the two-group K/V GEMMs)
        if (groups == 1) {
            const long long per = cus, tiles_n = a.N / 256;
            const long long T = (long long)((a.M + 255) / 256) * tiles_n, TH = (long long)((a.M + 127) / 128) * tiles_n;
            auto rounds = [&](long long tiles) { return (tiles + per - 1) / per; };
            // (half-tile time over full-tile time: 18.3 / 29 us measured at K = 1024 — the fixed per-tile cost halves with
            // the tile —, 0.75 for the long K loops; profiles/README.md)
            const double half = a.K <= 1024 ? 0.63 : 0.75, launch = 8.0 / (8.4 + 1.47 * (a.K / BK));
            const double cost_a = (double)rounds(T), cost_c = half * (double)rounds(TH);
            double cost_b = 1e30;
            long long head_rows = 0;
            if (T > per && per % tiles_n == 0) {
                const long long full = T / per;
                head_rows = full * per / tiles_n * 256;
                const long long tail_half = (long long)((a.M - head_rows + 127) / 128) * tiles_n;
                if (head_rows < a.M) cost_b = (double)full + half * (double)rounds(tail_half) + launch;
            }
            // (d) all 192 x 256 tiles (round 4): three quarters of a full tile's MFMA work and stores, but its two 8-MFMA phases do
            // not cover their partner's memory segments — measured 0.85 of a full tile's time (0.87 at K <= 1024; profiles/
            // r04t_t192_ab.json).  A 32-image shard's first layer is 2.25 rounds of full tiles and exactly 3 of these, its mlp
            // launches 1.125 and 1.5.  Plain launches only (no training epilogue, no statistics: the kernel is built for XMODE 0).
            const bool plain = !(a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD | TP_LINEAR_NO_STORE | TP_LINEAR_ROW_STATS)) &&
                               !a.acc_init && !a.attn_mode && !a.stats_parts && !a.tri && a.tt_rows == 0;   // (K-major launches: never)
            const long long T192 = (long long)((a.M + 191) / 192) * tiles_n;
            const double cost_d = plain ? (a.K <= 1024 ? 0.87 : 0.85) * (double)rounds(T192) : 1e30;
            const double best_abc = cost_c < cost_a - 0.05 && cost_c <= cost_b && TH >= 64 ? cost_c : (cost_b < cost_a - 0.05 ? cost_b : cost_a);
            if (tile_knob != 4 && T192 >= 64 && cost_d < best_abc - 0.05) return ROUTE_G8_192;
            if (TH >= 64 && cost_c < cost_a - 0.05 && cost_c <= cost_b) return ROUTE_G8_HALF;
            if (cost_b < cost_a - 0.05) { *head_rows_out = head_rows; return ROUTE_G8_SPLIT; }
        }
    }
    // TP_TUNE_GEMM_TILE = 3 (tests, A/Bs): every tile of a plain launch a 192 x 256 tile
    if (tuning(TP_TUNE_GEMM_TILE) == 3 && a.tile == 0 && !a.half_tiles && !a.A_parts[0] && a.N % 256 == 0 && a.m_begin == 0 && a.m_end == 0 &&
        !(a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD | TP_LINEAR_NO_STORE | TP_LINEAR_ROW_STATS)) && !a.acc_init && !a.attn_mode &&
        !a.stats_parts && !a.tri && a.tt_rows == 0)
        return ROUTE_G8_192;
    // TP_TUNE_GEMM_TILE = 2 (tests, A/Bs): every tile of the ping-pong kernel a 128 x 256 half tile
    if (tuning(TP_TUNE_GEMM_TILE) == 2 && a.tile == 0 && !a.half_tiles && !a.A_parts[0] && a.N % 256 == 0 && a.K >= 2 * BK &&
        a.m_begin == 0 && a.m_end == 0)
        return ROUTE_G8_HALF;
    if ((a.half_tiles && a.N % 256 == 0) || gemm_pick_tile(a.M, a.N, a.tile, a.groups) == 256) return ROUTE_G8;
    return ROUTE_SMALL;
}

// Whether gemm_launch sends `a` to the pair kernel (tp_gemm_pair.hip).  Its 256 x 128 tiles move 1.5 x the operand bytes of the
// 256 x 256 ping-pong tile through the CU's L1 -> LDS path, which is what bounds it (profiles/r04d_pair_probe.json: 1.58 us of memory-side
// work per K-tile with NO MFMAs against 1.14 us of matrix work) — on full-chip launches it loses 12-17 %.  Where it measured faster
// (or equal) is the short-K launch of 1.5 .. 2.5 rounds of its 512 workgroups, where the ping-pong kernel's 256 workgroups run 1.125
// or 2.25 rounds and every tile's epilogue is exposed: the query-side GEMMs at B = 256, the statistics / K / V / mlp[0] launches of
// a 32 .. 64-image shard (profiles/r04b_pair_ab.json).  TP_TUNE_PAIR_GEMM: 0 that policy | 1 never | 2 wherever supported.
static bool gemm_takes_pair_route(int in_dtype, int out_dtype, const GemmArgs& a) {
    const int mode = tuning(TP_TUNE_PAIR_GEMM);
    if (a.a_k_dup) {
        // the absorbed schedule's per-head V GEMM (N = 128 per head: the 256-column kernel cannot take it): the pair kernel from half a
        // round of its tiles on, the 128-tile kernel below that and with TP_TUNE_PAIR_GEMM = 1 — same K order, same bits either way
        const long long t = (long long)((a.M + 255) / 256) * (a.N / 128) * (a.groups > 0 ? a.groups : 1);
        return mode != 1 && a.tile == 0 && t * 2 >= gemm_pair_workgroups() && gemm_pair_supports(in_dtype, out_dtype, a);
    }
    if (mode == 1 || a.tile != 0 || (tuning(TP_TUNE_GEMM_TILE) != 0 && tuning(TP_TUNE_GEMM_TILE) != 4)) return false;
    if (!gemm_pair_supports(in_dtype, out_dtype, a)) return false;
    if (mode == 2) return true;
    const long long tiles = (long long)((a.M + 255) / 256) * (a.N / 128) * (a.groups > 0 ? a.groups : 1);
    const long long wgs = gemm_pair_workgroups();
    return a.K <= 1024 && tiles * 2 >= 3 * wgs && tiles * 2 < 5 * wgs;
}

// Where gemm_launch would send a plain K-contiguous launch of this shape under the tuning of the moment (test hook: the routing
// POLICY is host logic and is pinned by CPU tests): 0 128-tile kernel | 1 full 256 x 256 tiles | 2 all half tiles | 3 full rounds
// + a half-tile tail launch | 4 192 x 256 tiles | 5 the pair kernel
int gemm_route_of(int in_dtype, int out_dtype, const GemmArgs& a) {
    if (gemm_takes_pair_route(in_dtype, out_dtype, a)) return 5;
    long long h = 0;
    return gemm_route(a, &h);
}

bool gemm_uses_small_kernel(const GemmArgs& a) {
    long long h = 0;
    // (the LayerNorm-fold launches that ask are fp16 in / fp16 out)
    if (a.tt_rows == 0 && !a.stats_parts && gemm_takes_pair_route(TP_F16, TP_F16, a)) return false;
    return a.tt_rows == 0 && gemm_route(a, &h) == ROUTE_SMALL;
}

int gemm_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.N % 128 != 0 || a.K % BK != 0) {
        set_error("tp gemm: unsupported shape M=%d N=%d K=%d (need N%%128==0, K%%64==0)", a.M, a.N, a.K);
        return TP_ERR_INVALID_ARG;
    }
    {   // the epilogue stores through a 32-bit-ranged buffer descriptor (tp_gemm_common.h)
        const long long out_bytes = (long long)a.M * a.ldc * (out_dtype == TP_F32 ? 4 : 2);
        // (+ one tile of rows: the offsets of the dropped rows past M must not wrap around 32 bits either)
        if ((long long)(a.M + 256) * a.ldc * (out_dtype == TP_F32 ? 4 : 2) >= (1ll << 32)) {
            set_error("tp gemm: output of %lld bytes per group exceeds the 4 GiB a single launch can address "
                      "(M=%d ldc=%lld): split the rows over several calls", out_bytes, a.M, (long long)a.ldc);
            return TP_ERR_INVALID_ARG;
        }
    }
    if (a.attn_mode) {                                  // region attention in the epilogue (tp_gemm_common.h)
        if ((a.attn_mode != 1 && a.attn_mode != 2) || in_dtype != TP_F16 || out_dtype != TP_F16 || !a.acc_init || !a.bias ||
            !a.colsum || (!a.stats_in && !a.stats_parts) || a.flags != TP_LINEAR_LN_FOLD || a.groups != 1 || a.M % 8 != 0 || a.K != 16 * BK ||
            !a.attn_logits || (a.attn_mode == 1 ? (!a.attn_q || (a.attn_ldq_bytes & 15)) : !a.C)) {
            set_error("tp gemm: attn_mode needs fp16 operands, the LayerNorm-fold operands with acc_init, one group, K = 1024, M %% 8 == 0");
            return TP_ERR_INVALID_ARG;
        }
    }
    if ((a.flags & TP_LINEAR_ROW_STATS) && out_dtype == TP_F32) {
        set_error("tp gemm: ROW_STATS with fp32 output is not supported");
        return TP_ERR_INVALID_ARG;
    }
    if (a.tt_rows > 0) {                                // K-major operands: the ping-pong kernel only
        if (a.N % 256 != 0 || (a.lda_bytes & 15) || (a.ldw_bytes & 15) || a.ldw_bytes == 0 || a.flags != 0) {
            set_error("tp gemm: K-major operands need N %% 256 == 0, 16-byte aligned row strides and no epilogue flags");
            return TP_ERR_INVALID_ARG;
        }
        return gemm8_launch(in_dtype, out_dtype, a, stream);
    }
    if (a.m_begin < 0 || a.m_begin % 128 != 0 || (a.m_begin > 0 && a.m_begin >= a.M) ||
        (a.m_end != 0 && (a.m_end <= a.m_begin || a.m_end > a.M))) {
        set_error("tp gemm: bad row window [%d, %d) of %d rows", a.m_begin, a.m_end, a.M);
        return TP_ERR_INVALID_ARG;
    }
    // (a third main loop — one wave per SIMD, 128 x 128 wave tiles, AGPR accumulators, one barrier per K-tile — was built,
    // measured 8-22 % slower and removed in round 4: commit 638f1b9, profiles/r04k_solo_gemm_ab.json)
    if (gemm_takes_pair_route(in_dtype, out_dtype, a)) return gemm_pair_launch(in_dtype, out_dtype, a, stream);
    if (a.a_k_dup) {                                    // ... otherwise the 128-tile kernel (contiguous A, no training epilogue)
        const bool strided = a.rows_per_batch < a.M || a.a_region_s > 0;
        if (a.a_k_dup % BK != 0 || 2 * a.a_k_dup > a.K || strided || a.A_parts[0] || a.tri || a.tt_rows || (a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD))) {
            set_error("tp gemm: a_k_dup needs a contiguous A, K >= 2 a_k_dup, a_k_dup %% 64 == 0 and an inference epilogue");
            return TP_ERR_INVALID_ARG;
        }
        GemmArgs s = a; s.tile = 128;
        if (in_dtype == TP_F16 && out_dtype == TP_F16) return launch_types<f16_t, f16_t>(s, stream);
        if (in_dtype == TP_F16 && out_dtype == TP_F32) return launch_types<f16_t, float>(s, stream);
        if (in_dtype == TP_BF16 && out_dtype == TP_BF16) return launch_types<bf16_t, bf16_t>(s, stream);
        if (in_dtype == TP_BF16 && out_dtype == TP_F16) return launch_types<bf16_t, f16_t>(s, stream);
        set_error("tp gemm: a_k_dup: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
        return TP_ERR_INVALID_ARG;
    }
    {
        long long head_rows = 0;
        const int route = gemm_route(a, &head_rows);
        if (a.stats_parts && route != ROUTE_SMALL) {
            set_error("tp gemm: stats_parts (in-kernel LayerNorm merge) is served by the 128-tile kernel only (see gemm_uses_small_kernel)");
            return TP_ERR_INVALID_ARG;
        }
        if (route == ROUTE_G8_192) {
            GemmArgs h = a; h.tile192 = 1;
            return gemm8_launch(in_dtype, out_dtype, h, stream);
        }
        if (route == ROUTE_G8_HALF) {
            GemmArgs h = a; h.half_tiles = 1;
            return gemm8_launch(in_dtype, out_dtype, h, stream);
        }
        if (route == ROUTE_G8_SPLIT) {
            GemmArgs head = a, rest = a;
            head.m_end = (int)head_rows;
            rest.m_begin = (int)head_rows; rest.half_tiles = 1; rest.tile_counters = nullptr;
            if (int rc = gemm8_launch(in_dtype, out_dtype, head, stream)) return rc;
            return gemm8_launch(in_dtype, out_dtype, rest, stream);
        }
        if (route == ROUTE_G8) return gemm8_launch(in_dtype, out_dtype, a, stream);
    }
    if (in_dtype == TP_BF16) {
        if (out_dtype == TP_BF16) return launch_types<bf16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch_types<bf16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch_types<bf16_t, float>(a, stream);
    } else if (in_dtype == TP_F16) {
        if (out_dtype == TP_BF16) return launch_types<f16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch_types<f16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch_types<f16_t, float>(a, stream);
    }
    set_error("tp gemm: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
    return TP_ERR_INVALID_ARG;
}

}  // namespace tp
