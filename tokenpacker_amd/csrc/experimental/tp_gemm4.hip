// (round 6: FETCH = 2 / 3 — the SPREAD schedule re-built from the disassembly of the vendor's hand-written kernel, see G4Sched below and DESIGN.md 5.9.)
// EXPERIMENTAL (round 5; NOT part of libtokenpacker_hip.so — `make exp` links it into libtokenpacker_exp.so for tools/solo_ab.py).
// Round 4's one-wave-per-SIMD kernel (commit 638f1b9) with a second operand fetch: FETCH = 1 stages the loop's operands through
// REGISTERS (buffer_load_dwordx4 -> VGPR -> ds_write_b128, the vendor library's way — its kernel for these shapes is a hand-written
// MT256x256x64, MI16x16, 4-wave stream-K kernel, profiles/r05m) instead of LDS-DMA.  Why that could matter HERE although the two
// paths have the same throughput in isolation (tools/probes/operand_fetch_probe.hip): an LDS-DMA instruction holds the issuing
// wave's in-order stream for ~33 cycles (r04k), which a lone wave per SIMD pays in MFMA slots; a register load does not.
//
// tp_gemm4.hip — the 256x256x64 "solo" MFMA kernel: ONE wave per SIMD (4 waves, 512 registers each), wave tile 128 x 128, the
// matrix pipe fed by a single statically interleaved instruction stream instead of by two waves taking turns.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]^T ),  N % 256 == 0, K % 128 == 0   (reference builder.py:112,113,120,126-136)
//
// Why a third main loop beside tp_gemm8.hip (8 waves, ping-pong) and tp_gemm_pair.hip (two 4-wave workgroups per CU): the
// ping-pong loop hands the pipe from one wave to its partner across a barrier every 16 MFMAs, and each hand-over leaves it idle
// for a few tens of cycles — the loop runs at 1.43–1.47 us per K-tile against 1.14 us of MFMA work.  Here a wave
// never hands the pipe over: its MFMAs of k-step s are interleaved, four at a time, with the ds_read_b128 of the fragments of
// k-step s + 1 and the LDS-DMA of K-tile t + 2, in program order pinned with sched_barrier — ONE workgroup barrier per K-tile
// (64 KiB of operands, 128 MFMAs per wave) instead of eight.
//
//   * 4 waves (2 along M x 2 along N), wave tile 128 x 128 = 8 x 8 accumulator fragments = 256 fp32 registers (AGPRs); the
//     operand fragments of one 32-k step are 8 + 8 ds_read_b128 = 64 registers, double buffered.  LDS traffic per MFMA is half
//     of the 128 x 64 wave tile's (16 reads per 64 MFMAs).
//   * LDS: two K-tile buffers of 64 KiB — A rows 0..255 | W rows 0..255, 128 B (64 k) per row, lane-linear DMA image with the
//     `slot ^= row & 7` swizzle on the source address and on the fragment read (tp_gemm8.hip).
//   * K-tile t (buffer t & 1), k-steps (t,0) (t,1):
//         step (t,0): MFMAs on fragment set 0; reads fragment set 1 <- buffer t, k-half 1
//         s_waitcnt vmcnt(0) — K-tile t + 1 has landed (this wave's pieces) —, lgkmcnt(0), s_barrier          [B_t]
//         step (t,1): MFMAs on set 1; reads set 0 <- buffer t + 1, k-half 0; issues the 16 DMA pieces of K-tile t + 2 -> buffer t
//     RAW: K-tile t + 1 is first read behind B_t, which every wave passes after its own pieces have landed.
//     WAR: buffer t is last read by the reads issued in step (t,0), retired at the lgkmcnt(0) in front of B_t; its re-fill is
//          issued behind B_t.  A piece is in flight for one K-tile (~1.1 us) before it is waited for.
//   * persistent, per-XCD tile queues, the next tile's prologue (K-tiles 0 and 1) issued before the epilogue, epilogue
//     parameters by LDS-DMA, gemm_epilogue in its LAZY_PAR form — as in tp_gemm_pair.hip.  Same fragment layout, MFMA order
//     per accumulator (k ascending) and epilogue arithmetic as the other kernels: bit-identical results.
#include "../tp_gemm_common.h"
#include <atomic>
#include <mutex>

namespace tp {

namespace {

// The MFMA with its accumulator PINNED to AGPRs (inline asm, "+a"): with 256 accumulator registers the register allocator's own
// choice (VGPR-form MFMAs rewritten to AGPR form after allocation) left dst != srcC on most of them and ~2.7 v_accvgpr moves
// per MFMA inside the K loop.  The asm is volatile: the stream keeps the order it is written in.  What the hazard recogniser no
// longer sees: an accumulator is re-read by an MFMA 64 MFMAs later (hardware interlocked) and by the epilogue's VALU only
// behind the loop's closing barrier and the next tile's prologue issue (>> the 18 wait states of the longest XDL -> VALU rule;
// two s_nop 7 are placed there anyway).
template <typename T> struct Mma4;
// run_guarded (the spread schedule's tail K-tiles): outside the steady loop the register allocator re-homes accumulators (v_accvgpr_mov a248, a0
// two instructions in front of the MFMA that reads a[248:251] — found as ONE stale register per 16 x 16 block, tools/probes/solo_debug.py);
// the hazard recogniser cannot place the wait states for an instruction it does not see, so the asm carries them itself.
template <> struct Mma4<bf16_t> {
    static __device__ __forceinline__ void run(bf16x8 a, bf16x8 b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void run_guarded(bf16x8 a, bf16x8 b, f32x4& c) {
        asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
};
template <> struct Mma4<f16_t> {
    static __device__ __forceinline__ void run(f16x8 a, f16x8 b, f32x4& c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void run_guarded(f16x8 a, f16x8 b, f32x4& c) {
        asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
};

// ---- FETCH >= 2: the SPREAD schedule (round 6) --------------------------------------------------------------------------------
// One K-tile = 128 MFMAs in program order m = 64 s + 8 j + i (k-step s, W fragment j, A fragment i); behind MFMA m at most ONE other
// instruction (a fragment read, an LDS-DMA piece, or a wait + barrier), never two memory instructions in a row, the DMA pieces of
// K-tile t + 2 spread over the whole iteration instead of riding the second k-step only, and THREE barriers per K-tile so that each
// operand's region of buffer t & 1 is re-filled as soon as its last read has retired:
//     m  0..14   reads A set 1 (buffer B, k-half 1)                     -> lgkmcnt(0), barrier #1: A region of buffer B is free
//     m 22..45   DMA A pieces 0..7 of K-tile t + 2 -> buffer B  |  reads W set 1   -> lgkmcnt(0), barrier #2: W region free
//     m 52..80   DMA W pieces 0..7
//     m 84       vmcnt(16) — the 16 pieces issued in iteration t - 1 (K-tile t + 1) have landed —, barrier #3
//     m 86..116  reads set 0 <- buffer B ^ 1, k-half 0                   -> lgkmcnt(0) behind MFMA 127
// A piece is in flight for 1.0 .. 1.5 K-tile periods before it is waited for (schedule 0: 0.5 .. 1.0, behind a vmcnt(0)).  PH = 1
// (odd waves when FETCH = 3) moves every memory instruction one MFMA slot later, so the four SIMDs of a CU do not present their reads
// and DMA pieces to the LDS / the texture addresser in the same cycles.
enum { G4_NONE = 0, G4_RA1, G4_RW1, G4_RA0, G4_RW0, G4_DA, G4_DW, G4_LB, G4_VB, G4_L };
struct G4Slot { int kind, arg; };
struct G4Sched { G4Slot s[128]; };
constexpr G4Sched g4_make_sched(int ph) {
    G4Sched r{};
    for (int i = 0; i < 8; ++i) r.s[0 + 2 * i + ph] = G4Slot{G4_RA1, i};
    r.s[21 + ph] = G4Slot{G4_LB, 0};
    for (int q = 0; q < 8; ++q) r.s[22 + 3 * q + ph + (ph ? 0 : 0)] = G4Slot{G4_DA, q};
    for (int j = 0; j < 8; ++j) r.s[24 + 3 * j + ph] = G4Slot{G4_RW1, j};
    r.s[50 + ph] = G4Slot{G4_LB, 1};
    for (int q = 0; q < 8; ++q) r.s[52 + 4 * q + ph] = G4Slot{G4_DW, q};
    r.s[83 + ph] = G4Slot{G4_VB, 16};
    for (int i = 0; i < 8; ++i) r.s[86 + 2 * i + ph] = G4Slot{G4_RA0, i};
    for (int j = 0; j < 8; ++j) r.s[102 + 2 * j + ph] = G4Slot{G4_RW0, j};
    r.s[127] = G4Slot{G4_L, 0};
    return r;
}
template <int PH> struct G4SchedOf { static constexpr G4Sched value = g4_make_sched(PH); };
template <int I, int N, typename F> __device__ __forceinline__ void g4_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g4_static_for<I + 1, N>(f); }
}

constexpr int G4_BM = 256, G4_BN = 256, G4_WM = 128, G4_WN = 128;
constexpr int G4_HALF = 256 * ROW_BYTES;           // 32 KiB: the A (or W) rows of one K-tile
constexpr int G4_BUF = 2 * G4_HALF;                // 64 KiB: one K-tile
constexpr int G4_L_PAR = 2 * G4_BUF;               // bias[256] | colsum[256] | (mean, rstd)[256]  (gemm_epilogue's LDS_PARAMS layout)
constexpr int G4_L_RED = G4_L_PAR + 4096;          // row-statistics scratch [BN / 64][BM][2] floats
constexpr int G4_L_NEXT = G4_L_RED + 8192;
constexpr int G4_LDS_BYTES = G4_L_NEXT + 16;
static_assert(G4_LDS_BYTES <= 160 * 1024, "LDS budget of a CU");

}  // namespace

// AMODE: 0 = A rows contiguous (lda), 1 = rows in batches with a batch stride, optionally region-major (tp_gemm8.hip)
// DBG (TP_TUNE_PAIR_DEBUG, probe builds — garbage results): 1 no DMA in the loop | 2 no fragment reads | 4 no MFMAs
template <typename TI, typename TO, int AMODE, int DBG = 0, int FETCH = 0>
__global__ void __launch_bounds__(256)
gemm4_kernel(const GemmArgs p, const int tiles_m, const int tiles_n) {
    using X8 = typename Vec<TI>::x8;
    constexpr int BM = G4_BM, BN = G4_BN, WM = G4_WM, WN = G4_WN;
    constexpr int FM = WM / 16, FN = WN / 16;           // 8 x 8 accumulator fragments per wave
    constexpr int L_PAR = G4_L_PAR, L_NEXT = G4_L_NEXT;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = blockIdx.y;
    const int nk = p.K / BK;                            // even (checked on the host)
    const long long ldw = p.ldw_bytes ? p.ldw_bytes : (long long)p.K * 2;

    // ---- this workgroup's tile list (tp_gemm8.hip) ----
    const int ntiles = tiles_m * tiles_n, nwg = gridDim.x;
    int L, L_end, L_step;
    int* queue = nullptr;
    int queue_base = 0;
    if (ntiles <= nwg) {
        L = xcd_remap(blockIdx.x, nwg); L_end = L + 1; L_step = 1;
    } else if ((nwg & 7) == 0) {
        const int x = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = x * q + (x < r ? x : r);
        L = start + (blockIdx.x >> 3); L_end = start + q + (x < r ? 1 : 0); L_step = nwg >> 3;
        if (p.tile_counters) { queue = p.tile_counters + g * 8 + x; queue_base = start + L_step; }
    } else {
        L = blockIdx.x; L_end = ntiles; L_step = nwg;
    }
    if (L >= L_end) return;

    // ---- per-tile state ----
    int m0, n0, tile_n;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
    int voff_a[8], voff_w[8];                           // per-lane DMA source offsets of this wave's 8 + 8 pieces of a K-tile
    const int kslot = (lane & 7) ^ (lane >> 3);
    auto a_row_off = [&](int row) __attribute__((always_inline)) -> long long {
        if constexpr (AMODE != 0) {
            if (p.a_region_s > 0) row = region_major_to_raster(row, p.a_region_g, p.a_region_s);
            const int b = row / p.rows_per_batch;
            const int t = row - b * p.rows_per_batch;
            return (long long)b * p.a_batch_stride_bytes + (long long)t * p.lda_bytes;
        } else {
            return (long long)row * p.lda_bytes;
        }
    };
    // piece q of wave w = tile rows 64 w + 8 q .. + 7 (1 KiB): lane -> row 64 w + 8 q + lane / 8, 16-B slot' = lane % 8 holding
    // logical k-slot (lane % 8) ^ (lane / 8)
    auto setup_tile = [&](const int Lt) __attribute__((always_inline)) {
        const int tm = Lt / tiles_n;
        tile_n = Lt - tm * tiles_n;
        m0 = tm * BM; n0 = tile_n * BN;
        rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + g * p.w_gs + (long long)n0 * ldw), 0, 0x7fffffff, 0x00020000);
        long long a_tile_off = a_row_off(m0);
        if constexpr (AMODE != 0)
            if (p.a_region_s > 0) a_tile_off = (long long)(m0 / p.rows_per_batch) * p.a_batch_stride_bytes;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + g * p.a_gs + a_tile_off), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 64 * wave + 8 * q + (lane >> 3);
            int row = m0 + r;
            row = row < p.M ? row : p.M - 1;
            if constexpr (AMODE == 0) voff_a[q] = __mul24(row - m0, (int)p.lda_bytes) + kslot * 16;
            else voff_a[q] = (int)(a_row_off(row) - a_tile_off) + kslot * 16;
            voff_w[q] = __mul24(r, (int)ldw) + kslot * 16;
            // FETCH 4 / 5 (TIMING PROBES, garbage results — round 6): the addresses a TILE-MAJOR operand layout would produce (each 256-row x 64-k
            // slab 32 KiB contiguous: a DMA piece is 1 KiB contiguous instead of 8 rows x 128 B a row stride apart), inside the tile's own
            // 256 x K bytes.  4: both operands  5: W only (the weights are ours to lay out; A is the caller's)
            if constexpr (FETCH == 4) voff_a[q] = (row - m0) * ROW_BYTES + kslot * 16;
            if constexpr (FETCH == 4 || FETCH == 5) voff_w[q] = r * ROW_BYTES + kslot * 16;
        }
    };
    // piece idx (0..7: A, 8..15: W) of K-tile kt -> buffer `buf`
    auto issue_piece = [&](auto IDX_, const int buf, const int kt) __attribute__((always_inline)) {
        constexpr int idx = decltype(IDX_)::value;
        constexpr bool is_a = idx < 8;
        constexpr int q = idx & 7;
        char* dst = smem + buf * G4_BUF + (is_a ? 0 : G4_HALF) + (wave * 8 + q) * 1024;
        constexpr int KSTEP_A = FETCH == 4 ? 256 * ROW_BYTES : ROW_BYTES, KSTEP_W = (FETCH == 4 || FETCH == 5) ? 256 * ROW_BYTES : ROW_BYTES;
        // FETCH 6 / 7 (round 6): the cache-policy bits of the LDS-DMA loads — 6: nt (streaming, no allocation in the vector cache), 7: sc0
        constexpr int AUX = FETCH == 6 ? 2 : FETCH == 7 ? 1 : 0;
        if constexpr (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)dst, 16, voff_a[q], kt * KSTEP_A, 0, AUX);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)dst, 16, voff_w[q], kt * KSTEP_W, 0, AUX);
    };
    auto issue_ktile = [&](const int buf, const int kt) __attribute__((always_inline)) {
        issue_piece(std::integral_constant<int, 0>{}, buf, kt); issue_piece(std::integral_constant<int, 1>{}, buf, kt);
        issue_piece(std::integral_constant<int, 2>{}, buf, kt); issue_piece(std::integral_constant<int, 3>{}, buf, kt);
        issue_piece(std::integral_constant<int, 4>{}, buf, kt); issue_piece(std::integral_constant<int, 5>{}, buf, kt);
        issue_piece(std::integral_constant<int, 6>{}, buf, kt); issue_piece(std::integral_constant<int, 7>{}, buf, kt);
        issue_piece(std::integral_constant<int, 8>{}, buf, kt); issue_piece(std::integral_constant<int, 9>{}, buf, kt);
        issue_piece(std::integral_constant<int, 10>{}, buf, kt); issue_piece(std::integral_constant<int, 11>{}, buf, kt);
        issue_piece(std::integral_constant<int, 12>{}, buf, kt); issue_piece(std::integral_constant<int, 13>{}, buf, kt);
        issue_piece(std::integral_constant<int, 14>{}, buf, kt); issue_piece(std::integral_constant<int, 15>{}, buf, kt);
    };

    // ---- fragment reads: fragment i of A = tile rows wm * 128 + 16 i .. + 15, k-half ks; swizzled slot (row & 7 == lane & 7) ----
    const int slot0 = (((lane >> 4)) ^ (lane & 7)) << 4, slot1 = (((4 + (lane >> 4))) ^ (lane & 7)) << 4;
    const char* rd_a[2][2];     // [buffer][k-half]
    const char* rd_w[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        rd_a[b][0] = smem + b * G4_BUF + (wm * 128 + (lane & 15)) * ROW_BYTES + slot0;
        rd_a[b][1] = smem + b * G4_BUF + (wm * 128 + (lane & 15)) * ROW_BYTES + slot1;
        rd_w[b][0] = smem + b * G4_BUF + G4_HALF + (wn * 128 + (lane & 15)) * ROW_BYTES + slot0;
        rd_w[b][1] = smem + b * G4_BUF + G4_HALF + (wn * 128 + (lane & 15)) * ROW_BYTES + slot1;
    }

    f32x4 acc[FM][FN];
    X8 fa[2][8], fb[2][8];      // [fragment set][fragment]
    // FETCH = 1: the 16 pieces of ONE K-tile in registers (64 VGPRs): loaded while K-tile u's second k-step runs, written to LDS one
    // K-tile later (behind the barrier that frees the buffer), read by the MFMAs one K-tile after that — K-tile t + 2's data are loaded
    // in step (t - 1, 1) and written in step (t, 1), like the DMA pieces they replace are issued in step (t, 1) and waited for at B_(t+1)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 stg[16];
    auto load_piece = [&](auto IDX_, const int kt) __attribute__((always_inline)) {
        constexpr int idx = decltype(IDX_)::value;
        constexpr int q = idx & 7;
        if constexpr (idx < 8) stg[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[q], kt * ROW_BYTES, 0);
        else stg[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff_w[q], kt * ROW_BYTES, 0);
    };
    auto write_piece = [&](auto IDX_, const int buf) __attribute__((always_inline)) {
        constexpr int idx = decltype(IDX_)::value;
        constexpr int q = idx & 7;
        char* dst = smem + buf * G4_BUF + (idx < 8 ? 0 : G4_HALF) + (wave * 8 + q) * 1024 + lane * 16;
        *(u32x4*)dst = stg[idx];
    };
    if constexpr (DBG == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) { fa[s][i] = X8{}; fb[s][i] = X8{}; }
    }

    // One k-step: the 64 MFMAs of fragment set SET in 16 chunks of 4 (chunk c: W fragment c / 2, A fragments 4 (c % 2) .. + 3),
    // behind each chunk ONE read of the next set (c < 8: A fragment c, else W fragment c - 8) from (NB, NKS) when PRE, and ONE DMA
    // piece of K-tile kt_issue -> buffer IB when ISSUE.  sched_barrier pins the interleave.
    auto step = [&](auto SET_, auto PRE_, auto NB_, auto NKS_, auto ISSUE_, auto IB_, const int kt_issue) __attribute__((always_inline)) {
        constexpr int SET = decltype(SET_)::value, NB = decltype(NB_)::value, NKS = decltype(NKS_)::value, IB = decltype(IB_)::value;
        constexpr bool PRE = decltype(PRE_)::value, ISSUE = decltype(ISSUE_)::value;
        auto chunk = [&](auto C_) __attribute__((always_inline)) {
            constexpr int c = decltype(C_)::value;
            constexpr int j = c >> 1, i0 = (c & 1) * 4;
            if constexpr (DBG != 4) {
#pragma unroll
                for (int i = i0; i < i0 + 4; ++i) Mma4<TI>::run(fb[SET][j], fa[SET][i], acc[i][j]);
            } else {
                asm volatile("" :: "v"(fb[SET][j]), "v"(fa[SET][i0]), "v"(fa[SET][i0 + 1]), "v"(fa[SET][i0 + 2]), "v"(fa[SET][i0 + 3]));
            }
            if constexpr (PRE && DBG != 2) {
                if constexpr (c < 8) fa[SET ^ 1][c] = *(const X8*)(rd_a[NB][NKS] + c * 2048);
                else fb[SET ^ 1][c - 8] = *(const X8*)(rd_w[NB][NKS] + (c - 8) * 2048);
            }
            if constexpr (ISSUE && DBG != 1 && (FETCH == 0 || FETCH >= 4)) issue_piece(C_, IB, kt_issue);
            if constexpr (ISSUE && DBG != 1 && FETCH == 1) {
                write_piece(C_, IB);                                    // K-tile kt_issue (loaded a K-tile ago) -> the buffer B_t freed
                // ... and the registers take the K-tile after it (past the end: the last K-tile again — no branch in the stream;
                // those registers are never written to LDS)
                load_piece(C_, kt_issue + 1 < nk ? kt_issue + 1 : nk - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{}); chunk(std::integral_constant<int, 3>{});
        chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
        chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{});
        chunk(std::integral_constant<int, 8>{}); chunk(std::integral_constant<int, 9>{});
        chunk(std::integral_constant<int, 10>{}); chunk(std::integral_constant<int, 11>{});
        chunk(std::integral_constant<int, 12>{}); chunk(std::integral_constant<int, 13>{});
        chunk(std::integral_constant<int, 14>{}); chunk(std::integral_constant<int, 15>{});
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using T_ = std::true_type; using F_ = std::false_type;
    // K-tile in buffer B: step (t,0), the barrier B_t, step (t,1).  ISSUE: K-tile t + 2 exists; LAST: no K-tile t + 1
    auto ktile = [&](auto B_, auto ISSUE_, auto LAST_, const int t) __attribute__((always_inline)) {
        constexpr int B = decltype(B_)::value;
        constexpr bool LAST = decltype(LAST_)::value;
        step(I0{}, T_{}, B_, I1{}, F_{}, I0{}, 0);
        // (FETCH = 0: K-tile t + 1's DMA pieces must have landed.  FETCH = 1: nothing is in flight towards LDS except ds_writes, which the
        // lgkmcnt(0) below retires; the register loads of K-tile t + 2 stay in flight across the barrier)
        if constexpr (!LAST && (FETCH == 0 || FETCH >= 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): this wave's reads of buffer B (and its ds_writes) have retired
        if constexpr (!LAST) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        step(I1{}, std::integral_constant<bool, !LAST>{}, std::integral_constant<int, B ^ 1>{}, I0{}, ISSUE_, B_, t + 2);
    };

    // FETCH >= 2: one K-tile of the spread schedule (header).  PH: memory instructions one slot later
    auto ktile_spread = [&](auto B_, auto ISSUE_, auto LAST_, auto PH_, const int t) __attribute__((always_inline)) {
        constexpr int B = decltype(B_)::value, PH = decltype(PH_)::value;
        constexpr bool ISSUE = decltype(ISSUE_)::value, LAST = decltype(LAST_)::value;
        g4_static_for<0, 128>([&](auto M_) __attribute__((always_inline)) {
            constexpr int m = decltype(M_)::value;
            constexpr int ks = m >> 6, j = (m >> 3) & 7, i = m & 7;
            if constexpr (DBG == 4) asm volatile("" :: "v"(fb[ks][j]), "v"(fa[ks][i]));
            else if constexpr (ISSUE) Mma4<TI>::run(fb[ks][j], fa[ks][i], acc[i][j]);
            else Mma4<TI>::run_guarded(fb[ks][j], fa[ks][i], acc[i][j]);
            constexpr G4Slot op = G4SchedOf<PH>::value.s[m];
            if constexpr (op.kind == G4_RA1 && DBG != 2) fa[1][op.arg] = *(const X8*)(rd_a[B][1] + op.arg * 2048);
            if constexpr (op.kind == G4_RW1 && DBG != 2) fb[1][op.arg] = *(const X8*)(rd_w[B][1] + op.arg * 2048);
            if constexpr (op.kind == G4_RA0 && DBG != 2 && !LAST) fa[0][op.arg] = *(const X8*)(rd_a[B ^ 1][0] + op.arg * 2048);
            if constexpr (op.kind == G4_RW0 && DBG != 2 && !LAST) fb[0][op.arg] = *(const X8*)(rd_w[B ^ 1][0] + op.arg * 2048);
            if constexpr (op.kind == G4_DA && ISSUE && DBG != 1) issue_piece(std::integral_constant<int, op.arg>{}, B, t + 2);
            if constexpr (op.kind == G4_DW && ISSUE && DBG != 1) issue_piece(std::integral_constant<int, 8 + op.arg>{}, B, t + 2);
            if constexpr (op.kind == G4_LB) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if constexpr (!LAST) __builtin_amdgcn_s_barrier();
            }
            if constexpr (op.kind == G4_VB && !LAST) {
#ifdef TP_G4_VM0
                if constexpr (ISSUE && DBG != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                if constexpr (ISSUE && DBG != 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (op.kind == G4_L) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                // the hazard recogniser does not see the inline-asm MFMAs (Mma4): at the exit of the steady loop the register allocator
                // re-homes accumulators with v_accvgpr moves, which would read a[252:255] inside the last MFMA's latency (found as a
                // deterministic loss of exactly that MFMA's contribution, tools/probes/solo_debug.py)
                asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto k_loop_spread = [&](auto PH_) __attribute__((always_inline)) {
        int t = 0;
        for (; t < nk - 2; t += 2) { ktile_spread(I0{}, T_{}, F_{}, PH_, t); ktile_spread(I1{}, T_{}, F_{}, PH_, t + 1); }
        ktile_spread(I0{}, F_{}, F_{}, PH_, t); ktile_spread(I1{}, F_{}, T_{}, PH_, t + 1);
    };

    auto k_loop = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (DBG != 2) {                           // fragment set 0 <- buffer 0, k-half 0
#pragma unroll
            for (int i = 0; i < 8; ++i) { fa[0][i] = *(const X8*)(rd_a[0][0] + i * 2048); fb[0][i] = *(const X8*)(rd_w[0][0] + i * 2048); }
        }
        if constexpr (FETCH == 1 && DBG != 1) {             // K-tiles 0, 1 are in LDS (the prologue's DMA); K-tile 2 starts in registers
            if (nk > 2) {
                load_piece(std::integral_constant<int, 0>{}, 2); load_piece(std::integral_constant<int, 1>{}, 2);
                load_piece(std::integral_constant<int, 2>{}, 2); load_piece(std::integral_constant<int, 3>{}, 2);
                load_piece(std::integral_constant<int, 4>{}, 2); load_piece(std::integral_constant<int, 5>{}, 2);
                load_piece(std::integral_constant<int, 6>{}, 2); load_piece(std::integral_constant<int, 7>{}, 2);
                load_piece(std::integral_constant<int, 8>{}, 2); load_piece(std::integral_constant<int, 9>{}, 2);
                load_piece(std::integral_constant<int, 10>{}, 2); load_piece(std::integral_constant<int, 11>{}, 2);
                load_piece(std::integral_constant<int, 12>{}, 2); load_piece(std::integral_constant<int, 13>{}, 2);
                load_piece(std::integral_constant<int, 14>{}, 2); load_piece(std::integral_constant<int, 15>{}, 2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FETCH == 2) { k_loop_spread(I0{}); return; }
        if constexpr (FETCH == 3) {
            if (wave & 1) k_loop_spread(I1{}); else k_loop_spread(I0{});
            return;
        }
        int t = 0;
        for (; t < nk - 2; t += 2) { ktile(I0{}, T_{}, F_{}, t); ktile(I1{}, T_{}, F_{}, t + 1); }
        ktile(I0{}, F_{}, F_{}, t); ktile(I1{}, F_{}, T_{}, t + 1);
    };

    // ---- persistent: walk the tile list (tp_gemm_pair.hip) ----
    const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
    const float* __restrict__ colsum = (p.flags & TP_LINEAR_LN_FOLD) ? p.colsum + g * p.colsum_gs : nullptr;
    const float* __restrict__ stats_in = (p.flags & TP_LINEAR_LN_FOLD) ? p.stats_in + g * p.stats_in_gs : nullptr;
    auto dma_par = [&](const char* src, long long len, const int dst) __attribute__((always_inline)) {
        len = len < 0 ? 0 : (len > 0x7fffffff ? 0x7fffffff : len);
        const unsigned long long addr = (unsigned long long)src;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
        const int nr = __builtin_amdgcn_readfirstlane((int)len);
        const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nr, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + L_PAR + dst), 16, lane * 16, 0, 0, 0);
    };
    auto issue_params = [&](const int m0t, const int n0t) __attribute__((always_inline)) {
        const long long rows_left = (long long)p.M - m0t;
        if (wave == 0) { if (bias) dma_par((const char*)(bias + n0t), BN * 4, 0); }
        else if (wave == 1) { if (colsum) dma_par((const char*)(colsum + n0t), BN * 4, BN * 4); }
        else if (stats_in) {
            const int half = wave - 2;
            dma_par((const char*)(stats_in + (long long)(m0t + half * 128) * 2), (rows_left - half * 128) * 8, 2 * BN * 4 + half * 1024);
        }
    };
    if (!bias) ((float*)(smem + L_PAR))[tid] = 0.f;

    setup_tile(L);
    issue_ktile(0, 0); issue_ktile(1, 1);
    int drawn = 0;
    if (queue && tid == 0) drawn = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        if (queue && tid == 0) *(int*)(smem + L_NEXT) = drawn;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // K-tiles 0, 1 landed, earlier stores retired
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_params(m0, n0);
        k_loop();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");    // (XDL write -> VALU read of the accumulators: see Mma4)
        __builtin_amdgcn_s_barrier();                       // every wave's fragment reads have retired: both buffers are free
        __builtin_amdgcn_sched_barrier(0);
        const int m0c = m0, n0c = n0, tile_nc = tile_n;
        int Ln = L + L_step;
        if (queue) Ln = queue_base + __builtin_amdgcn_readfirstlane(*(const int*)(smem + L_NEXT));
        const bool has_next = Ln < L_end;
        float2 mean_rstd[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) mean_rstd[i] = make_float2(0.f, 1.f);
        if (has_next) {
            setup_tile(Ln);
            issue_ktile(0, 0); issue_ktile(1, 1);
            if (queue && tid == 0) drawn = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63;
        gemm_epilogue<TO, BM, BN, WM, WN, true, false, false, true>(acc, p, g, m0c, n0c, tile_nc, wm, wn, lane_e, tid_e, mean_rstd,
                                                                      smem + G4_L_RED, smem + L_PAR);
        if (!has_next) break;
        L = Ln;
        block_sync_lds();
    }
}

// ---- host side ------------------------------------------------------------------------------------------
#if !defined(TP_G4_PART) || TP_G4_PART == 0
bool gemm4_supports(int in_dtype, int out_dtype, const GemmArgs& a) {
    if (a.tt_rows > 0 || a.half_tiles || a.m_begin != 0 || a.m_end != 0 || a.stats_parts || a.parts_k_groups || a.A_parts[0]) return false;
    if (a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD | TP_LINEAR_NO_STORE)) return false;
    if (a.acc_init || a.attn_mode || a.tri) return false;
    if (a.N % G4_BN != 0 || a.K % (2 * BK) != 0 || a.K < 2 * BK) return false;
    if (a.lda_bytes >= (1 << 23) || (a.ldw_bytes ? a.ldw_bytes : (long long)a.K * 2) >= (1 << 23)) return false;
    if ((a.flags & TP_LINEAR_ROW_STATS) && out_dtype == TP_F32) return false;
    return in_dtype == TP_BF16 || in_dtype == TP_F16;
}

#endif
[[maybe_unused]] static int g4_fetch_mode = 0;       // set by tp_exp_gemm4 (single-threaded tool)
int gemm4_launch_spread(int in_dtype, int out_dtype, int dbg, bool stagger, const GemmArgs& a, hipStream_t stream);   // (-DTP_G4_PART=1 object)
template <typename TI, typename TO, int AMODE, int DBG = 0, int FETCH = -1>
static int launch4_cfg(const GemmArgs& a, hipStream_t stream) {
    if constexpr (FETCH < 0) {
        if constexpr (DBG == 0 && AMODE == 0 && std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value) {
            const int dbg = tuning(TP_TUNE_PAIR_DEBUG) & 7;
            if (dbg == 1 || dbg == 2 || dbg == 4) {
                const int f = g4_fetch_mode;
                if (f == 2) return gemm4_launch_spread(TP_F16, TP_F16, dbg, 0, a, stream);
                if (dbg == 1) return f == 1 ? launch4_cfg<TI, TO, AMODE, 1, 1>(a, stream) : launch4_cfg<TI, TO, AMODE, 1, 0>(a, stream);
                if (dbg == 2) return f == 1 ? launch4_cfg<TI, TO, AMODE, 2, 1>(a, stream) : launch4_cfg<TI, TO, AMODE, 2, 0>(a, stream);
                return f == 1 ? launch4_cfg<TI, TO, AMODE, 4, 1>(a, stream) : launch4_cfg<TI, TO, AMODE, 4, 0>(a, stream);
            }
        }
        // The spread schedule's 128-slot static_for is slow to compile, so its instantiations are a second object of this file
        // (-DTP_G4_PART=1, built in parallel: gemm4_launch_spread) and exist where tools/solo_ab.py uses them: contiguous A, half-precision
        // outputs; the SIMD-parity stagger (FETCH 3: measured, slower) only with -DTP_G4_STAGGER.
        if (g4_fetch_mode >= 4 && g4_fetch_mode <= 7) {
            if constexpr (AMODE == 0 && DBG == 0 && std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value)
                return g4_fetch_mode == 4 ? launch4_cfg<TI, TO, AMODE, 0, 4>(a, stream) : g4_fetch_mode == 5 ? launch4_cfg<TI, TO, AMODE, 0, 5>(a, stream)
                     : g4_fetch_mode == 6 ? launch4_cfg<TI, TO, AMODE, 0, 6>(a, stream) : launch4_cfg<TI, TO, AMODE, 0, 7>(a, stream);
            else { set_error("tp gemm4: the tile-major timing probes are built for contiguous fp16 -> fp16 launches"); return TP_ERR_INVALID_ARG; }
        }
        switch (g4_fetch_mode) {
            case 1: return launch4_cfg<TI, TO, AMODE, DBG, 1>(a, stream);
            case 2: case 3:
                if constexpr (AMODE == 0 && DBG == 0 && !std::is_same<TO, float>::value)
                    return gemm4_launch_spread(std::is_same<TI, bf16_t>::value ? TP_BF16 : TP_F16, std::is_same<TO, bf16_t>::value ? TP_BF16 : TP_F16, 0,
                                               g4_fetch_mode == 3, a, stream);
                else { set_error("tp gemm4: the spread schedule is built for contiguous A and half-precision outputs"); return TP_ERR_INVALID_ARG; }
            default: return launch4_cfg<TI, TO, AMODE, DBG, 0>(a, stream);
        }
    } else {
    auto kern = gemm4_kernel<TI, TO, AMODE, DBG, FETCH>;
    constexpr int lds = G4_LDS_BYTES;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    if (attr_err != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", lds, hipGetErrorString(attr_err));
        return TP_ERR_LAUNCH;
    }
    const int tiles_m = (a.M + G4_BM - 1) / G4_BM, tiles_n = a.N / G4_BN;
    const int ntiles = tiles_m * tiles_n;
    int nwg = ntiles;
    const int cap = gemm8_persistent_cus();
    if (nwg > cap && cap > 0) nwg = cap;
    dim3 grid((unsigned)nwg, (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, tiles_m, tiles_n);
    return check_launch("gemm4_kernel");
    }
}

#if defined(TP_G4_PART) && TP_G4_PART == 1
// the spread schedule's instantiations (dbg: the probe builds of the f16 -> f16 launch)
int gemm4_launch_spread(int in_dtype, int out_dtype, int dbg, bool stagger, const GemmArgs& a, hipStream_t stream) {
#ifdef TP_G4_STAGGER
    if (stagger && dbg == 0) {
        if (in_dtype == TP_BF16) return out_dtype == TP_BF16 ? launch4_cfg<bf16_t, bf16_t, 0, 0, 3>(a, stream) : launch4_cfg<bf16_t, f16_t, 0, 0, 3>(a, stream);
        return out_dtype == TP_BF16 ? launch4_cfg<f16_t, bf16_t, 0, 0, 3>(a, stream) : launch4_cfg<f16_t, f16_t, 0, 0, 3>(a, stream);
    }
#endif
    if (stagger) { set_error("tp gemm4: fetch 3 (spread schedule + SIMD-parity stagger) needs a -DTP_G4_STAGGER build"); return TP_ERR_INVALID_ARG; }
    if (dbg == 1) return launch4_cfg<f16_t, f16_t, 0, 1, 2>(a, stream);
    if (dbg == 2) return launch4_cfg<f16_t, f16_t, 0, 2, 2>(a, stream);
    if (dbg == 4) return launch4_cfg<f16_t, f16_t, 0, 4, 2>(a, stream);
    if (in_dtype == TP_BF16) return out_dtype == TP_BF16 ? launch4_cfg<bf16_t, bf16_t, 0, 0, 2>(a, stream) : launch4_cfg<bf16_t, f16_t, 0, 0, 2>(a, stream);
    return out_dtype == TP_BF16 ? launch4_cfg<f16_t, bf16_t, 0, 0, 2>(a, stream) : launch4_cfg<f16_t, f16_t, 0, 0, 2>(a, stream);
}
}  // namespace tp
#else
template <typename TI, typename TO>
static int launch4_types(const GemmArgs& a, hipStream_t stream) {
    const bool strided_a = a.rows_per_batch < a.M || a.a_region_s > 0;
    return strided_a ? launch4_cfg<TI, TO, 1>(a, stream) : launch4_cfg<TI, TO, 0>(a, stream);
}

int gemm4_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream) {
    if (!gemm4_supports(in_dtype, out_dtype, a)) {
        set_error("tp gemm4: launch not supported by the solo kernel (M=%d N=%d K=%d flags=%d)", a.M, a.N, a.K, a.flags);
        return TP_ERR_INVALID_ARG;
    }
#ifdef TP_G4_FAST      // (debug builds: one instantiation, seconds to compile)
    if (in_dtype == TP_F16 && out_dtype == TP_F32) return launch4_cfg<f16_t, float, 0, 0, TP_G4_FAST>(a, stream);
    set_error("tp gemm4: TP_G4_FAST build"); return TP_ERR_INVALID_ARG;
#else
    if (in_dtype == TP_BF16) {
        if (out_dtype == TP_BF16) return launch4_types<bf16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch4_types<bf16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch4_types<bf16_t, float>(a, stream);
    } else {
        if (out_dtype == TP_BF16) return launch4_types<f16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch4_types<f16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch4_types<f16_t, float>(a, stream);
    }
    set_error("tp gemm4: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
    return TP_ERR_INVALID_ARG;
#endif
}

}  // namespace tp

// tools/solo_ab.py: tp_linear's argument block on the experimental kernel.  fetch: 0 LDS-DMA | 1 register-staged | 2 LDS-DMA, spread
// schedule | 3 the same, odd waves one slot later
extern "C" int tp_exp_gemm4(const tp_linear_args* a, void* stream, int fetch) {
    using namespace tp;
    GemmArgs g{};
    g.A = (const char*)a->A; g.W = (const char*)a->W; g.C = (char*)a->C;
    g.bias = a->bias; g.stats_in = a->row_mean_rstd; g.colsum = a->colsum; g.stats_out = a->row_stats_out;
    g.rows_per_batch = (a->rows_per_batch > 0 && a->rows_per_batch < a->M) ? a->rows_per_batch : a->M;
    g.a_batch_stride_bytes = a->a_batch_stride * 2; g.lda_bytes = a->lda * 2; g.ldc = a->ldc;
    g.M = a->M; g.N = a->N; g.K = a->K; g.flags = a->flags; g.groups = 1;
    g4_fetch_mode = fetch;
    return gemm4_launch(a->dtype, a->out_dtype, g, (hipStream_t)stream);
}
#endif      // TP_G4_PART
