// tp_gemm8.hip — the 256x256x64 "ping-pong" MFMA kernel that carries the large contractions of the
// TokenPacker path on gfx950 (MI355X):   C[M,N] = epilogue( A[M,K] · W[N,K]^T ),  N % 256 == 0.
//
// Same operands, fragment layout and fused epilogue as tp_gemm.hip (reference builder.py:112,113,120,
// 126-130,136); what differs is the main loop, built for how a CDNA4 CU actually issues:
//
//   * 8 waves (2 along M x 4 along N), one workgroup per CU, 128 KiB of LDS = 2 K-tile buffers.  A SIMD
//     hosts one wave of each M-half; the two halves run ONE BARRIER APART, so while one wave of a SIMD is
//     in its MFMA segment (16 x v_mfma_f32_16x16x32, s_setprio 1) its partner is in its memory segment
//     (ds_read_b128 fragment reads + the LDS-DMA issue for a later K-tile) — matrix beside memory on every
//     SIMD, every segment, by construction instead of by luck of the scheduler.
//   * A K-tile is consumed in 4 phases, one 64x32 quadrant of the wave's 128x64 output each, in the order
//     (a0,b0) (a0,b1) (a1,b1) (a1,b0): fragment reads per phase are 12 / 4 / 8 / 0 ds_read_b128.
//   * HBM/L2 -> LDS by buffer_load_dwordx4 ... lds (1 KiB per wave instruction) in FOUR 16-KiB groups per
//     K-tile ordered by when they are first read: G0 = A rows of quadrant a0 (both M-halves), G1 = W rows of
//     b0 (all four N-quarters), G2 = W rows of b1, G3 = A rows of a1.  One group is issued per phase (two DMA
//     instructions per wave), up to two K-tiles ahead, and retired with a COUNTED s_waitcnt vmcnt(2*DEPTH):
//     the queue is never drained inside the loop and the loads stay in flight across the barriers.
//   * K advances through the buffer instruction's scalar offset: no per-iteration address VALU at all.
//   * LDS image is lane-linear (a DMA constraint); the bank-conflict swizzle slot' = slot ^ (row & 7) is
//     applied on the per-lane SOURCE address and again on the ds_read_b128 address (same involution).
//
// Hazard bookkeeping (phase g = 4*tile + p; group 0 = waves of M-half 0, group 1 runs one barrier later):
//   RAW  a group whose covering vmcnt sits in phase g_w (before that phase's first barrier) may be read
//        from phase g_w + 1 on:  every wave has passed its wait before any reader passes the barrier
//        that opens its read segment.
//   WAR  a region last read in phase g_r may be re-targeted by a DMA issued in phase >= g_r + 2 (the late
//        group's reads retire at its lgkmcnt(0), one barrier before the early group's issue segment).
//   Schedule: phase p of tile t issues   p=0: G2(t+1)  p=1: G3(t+1)  p=2: G0(t+2)  p=3: G1(t+2)
//   (re-target distance 3,3,2,3 phases) and waits vmcnt(2*DEPTH), DEPTH = 3: the group issued 3 phases
//   ago has landed; it is first read 1..2 phases after that.  (DEPTH = 4 is the latest legal placement —
//   4..5 phases of flight time per group — and measured the same: DMA latency is not what bounds the loop.)
//
// PERSISTENT form (default): one workgroup per CU walks a strided list of output tiles.  When the K loop of a
// tile ends, the DMA prologue of the NEXT tile (6 groups) and the loads of its epilogue parameters are issued
// BEFORE the current tile's epilogue runs, so HBM/L2 latency and the workgroup launch gap disappear behind the
// epilogue's VALU work and stores.  Epilogue parameters (bias, colsum, LayerNorm mean/rstd) are staged through
// 4 KiB of LDS: no ordinary vector load is ever consumed while the LDS-DMA queue is busy (hipcc would answer
// with s_waitcnt vmcnt(0) and drain it).  Variants that did NOT pay on MI355X and were removed: 32-MFMA
// segments (same wall time, the kernel is power-limited: fewer cycles came back as lower clock) and releasing the
// partner wave a few MFMAs early (-10 %: matrix beside matrix on one SIMD); see profiles/r01c_gemm_bench.json.
#include "tp_gemm_common.h"
#include <mutex>

#ifndef TP_G8_STAGGER_NS
#define TP_G8_STAGGER_NS 0      // A/B build flag (round 6), see gemm8_kernel
#endif

namespace tp {

namespace {

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 1 || N == 2 || N == 3 || N == 4 || N == 5 || N == 6 || N == 8 || N == 10 || N == 11 || N == 12 || N == 13, "unsupported count");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
}

constexpr int G8_BM = 256, G8_BN = 256, G8_WN = 64;
constexpr int G8_GROUP = 128 * ROW_BYTES;          // 16 KiB: one DMA group (128 rows x 128 B)
constexpr int G8_BUF = 4 * G8_GROUP;               // 64 KiB: one K-tile (G0 | G1 | G2 | G3)
constexpr int G8_RING = 2 * G8_BUF;                // 128 KiB of K-tile buffers
// persistent kernel, behind the ring: bias[256] | colsum[256] | (mean, rstd)[256] | acc_init[256] (5 KiB), 8 KiB
// row-statistics scratch, the next tile index drawn from the queue.
// XMODE 3 / 4 (attention epilogues): the parameters arrive by LDS-DMA with the tile's prologue into one of TWO buffers (the
// next tile's land while this tile's are still read): bias | colsum | (mean, rstd)[BM] | acc_init | logits [2 heads][BM]
// (XMODE 4), and the scratch is what the logits reduction needs (XMODE 3: 16 B per tile row).
constexpr int G8_PAR = 5120;
constexpr int g8_par_bytes(bool half, int xmode) { return xmode >= 3 ? 2 * (half ? 5120 : 7168) : G8_PAR; }
constexpr int g8_red_bytes(bool half, int xmode) { return xmode == 3 ? (half ? 128 : 256) * 16 : (xmode == 4 ? 0 : 8192); }
constexpr int g8_lds_bytes(bool half, bool persist, int xmode) {
    return half ? 9 * 16384 + g8_par_bytes(true, xmode) + g8_red_bytes(true, xmode) + 256
                : (persist ? 8 * 16384 + g8_par_bytes(false, xmode) + g8_red_bytes(false, xmode) + 256 : 8 * 16384);
}
#ifndef TP_G8_DEPTH
#define TP_G8_DEPTH 3
#endif
constexpr int G8_DEPTH = TP_G8_DEPTH;              // schedule 1: DMA groups left in flight by the in-loop wait (3 or 4 are legal;
                                                   // 4 = the latest legal wait placement measured no faster: r01u)
// Schedule 2 (round 5; full 256-row tiles — HALF and T192 keep schedule 1) — BUILT, BIT-IDENTICAL, MEASURED SLOWER, NOT THE DEFAULT
// (profiles/r05b_lib_ab_s2_vs_s1.json, two builds dlopen'ed side by side, arms interleaved: kv_layer0 +1.2 %, mlp2 +1.1 %, mlp0 +2.7 %,
// K = 1024 plain +2.7 %; whole forward 4.222 -> 4.256 ms; schedule 1 with DEPTH = 4: +-0, r05b_lib_ab_s1d4_vs_s1.json).  More operand
// bytes in flight and a lighter phase 0 do not help: the operand fetch is not latency-bound, and four DMA
// instructions back to back in one memory segment stall their wave at issue longer than two here and two there.  Kept behind
// -DTP_G8_SCHED=2 (`make variant`) as the A/B partner.  The round-4 probes (profiles/r04m_loop_probe.json) put the
// loop's two memory-side costs in the memory segment the partner wave's 16 MFMAs (256 cycles) must cover: phase 0 carries 12 fragment
// reads AND two DMA issues (~300 cycles), phase 3 no read at all; and the operand fetch runs at what 48 KiB in flight per CU deliver.
// Schedule 2 moves phase 0's DMA issue into phase 3 of the previous K-tile and waits only where a group is first read:
//     phase p of K-tile t issues   p=0: -   p=1: G3(t+1)   p=2: G0(t+2)   p=3: G1(t+2), G2(t+2)
//     (re-target distances 3 / 2 / 3, 2 phases: G3 of buffer (t+1)&1 last read in p2(t-1); G0, G1 of buffer t&1 in p0(t); G2 in p1(t))
//     waits (instructions left in flight; issue order ... G0(t) | G1(t) G2(t) | G3(t) | G0(t+1) | G1(t+1) G2(t+1) | G3(t+1) ...):
//       p0(t): G2(t) is read in p1  -> younger: G3(t) G0(t+1) G1(t+1) G2(t+1)                = vmcnt(8)
//       p1(t): G3(t) is read in p2  -> younger: G0(t+1) G1(t+1) G2(t+1) G3(t+1)              = vmcnt(8)
//       p2(t): nothing new is read in p3                                                     = no wait
//       p3(t): G0(t+1), G1(t+1) are read in p0(t+1) -> younger: G2(t+1) G3(t+1) G0(t+2) G1(t+2) G2(t+2) = vmcnt(10)
//     i.e. four to five 16-KiB groups in flight (64 - 80 KiB per CU instead of 48) and memory segments of 12 reads | 4 reads + 2 DMA |
//     8 reads + 2 DMA | 4 DMA instead of 12 + 2 | 4 + 2 | 8 + 2 | 0 + 2.  Tails: K-tile nk-2 issues G3(nk-1) only (waits 8 / 8 / - / 4),
//     K-tile nk-1 nothing (2 / 0 / - / -).  The prologue carries G2(1) as well (7 groups) — except in the K launch (XMODE 3), whose queries
//     ride in that very region of the ring: there K-tile 0 issues G2(1) in its phase 0, as in schedule 1, and every later count is
//     the same (G2(1) is older than G3(1), as G2(t) is older than G3(t) in the steady state).  Same MFMA order, same bits.
#ifndef TP_G8_SCHED
#define TP_G8_SCHED 1
#endif
constexpr int G8_SCHED = TP_G8_SCHED;

}  // namespace

// AMODE: 0 = A rows contiguous (lda), 1 = rows in batches with a batch stride (the CLIP tower's [:,1:] slices),
//        2 = like 1 with K split over four source tensors (the tower's hidden states consumed without torch.cat),
//        3 = "TT": both operands K-MAJOR — A[k][m], W[k][n], the contraction index is the row of both (weight
//            gradients dW = dY^T · X straight from the row-major activations, no transposed copies).  A DMA
//            instruction moves 4 k-rows x 128 contiguous columns (whole cache lines, like the K-contiguous modes) as
//            eight [4 k][16 col] chunks of 128 B; ds_read_b64_tr_b16 turns one chunk per 16-lane group into the
//            MFMA operand layout (lane = column, 4 consecutive k per read, two reads per 32-k MFMA step — the same
//            k permutation on both operands, so the contraction is unchanged).  Chunk slots are XOR-ed with the
//            k-quad's parity so that the two lane groups an LDS cycle serves hit different bank halves.  Rows past
//            the end of the contraction range read as zero (descriptor range check, rebuilt per K-tile).
//        4 = A K-major as in 3, W K-contiguous as in 0 (a weight gradient whose activation operand had to be
//            transposed anyway — cast, LayerNorm applied — while dY is read in place)
// HALF: 128 x 256 output tiles (the tail round of a launch, small M).  The wave tile is 64 x 64 — the a0 quadrant alone —
// so a K-tile is three DMA groups (G0 = the 128 A rows, G1 / G2 = W halves b0 / b1) consumed in TWO phases (a0,b0) (a0,b1)
// of 16 MFMAs; the ring holds three K-tiles of 48 KiB.  Schedule: phase 0 of tile t issues G2(t+1), phase 1 issues G0(t+2)
// and G1(t+2) — re-target distance 3 phases, flight time 2..3 phases — and both wait vmcnt(6).
// XMODE: 0 plain | 1 statistics only (TP_LINEAR_NO_STORE: no output) | 2 accumulators pre-loaded from GemmArgs::acc_init —
// the two GEMMs of the fused LayerNorm chain; separate instantiations so that the common kernels' register budget is
// not taxed (one more live value in the persistent loop tipped them into scratch).
// XMODE 3 / 4: 2 + region attention in the epilogue (GemmArgs::attn_mode 1 / 2, tp_gemm_common.h), K = 1024 (nk = 16).
//   3 (the K launch) needs the tile's queries [BM / 4 regions][256 columns] fp16 = 32 KiB (HALF: 16 KiB) in LDS when the
//   K loop ends, and the ring is all there is.  During the LAST K-tile the buffer of K-tile nk-2 is idle: its G2 | G3 part
//   (HALF: its G0 part) takes the queries, issued in the two phases of that K-tile that have no group of their own to issue
//   (re-target distance 3 phases, HALF 2; the tail's waits leave them in flight).  The next tile's prologue must not land
//   there: K-tile 0 of the next tile goes to the buffer of K-tile nk-1 instead (`ring_base` flips per tile, nk even), K-tile 1's
//   G0 | G1 beside the queries; HALF: nk % 3 == 1 puts K-tile nk-2 in slot 2, which the prologue (slot 0, slot 1's G0 | G1)
//   does not touch.  After the prologue one counted wait (its 12 / 10 instructions stay in flight) + a barrier publish the
//   queries; the logits leave as 4 B per row and head, K itself is never stored.
//   4 (the V launch) stages the tile's logits with the other epilogue parameters.
//   Both take their epilogue parameters by LDS-DMA (one 1-KiB instruction per wave, issued with the prologue, two buffers)
//   instead of the register-carried prefetch of the other modes: nothing rides through the epilogue in registers a
//   compiler-inserted s_waitcnt could trip over, and the count of instructions behind the queries is the same in every wave.
// PROBE (TP_TUNE_PAIR_DEBUG >> 4, timing probes of the K loop on fp16 -> fp16 plain launches, tools/loop_probe.py; results are
// GARBAGE): bit 0 no b0 fragment reads (phase 0) | 1 no b1 reads (phase 1) | 2 no a0 reads (phase 0) | 3 no a1 reads (phase 2) |
// 4 no DMA in the loop | 6 no MFMAs
// T192 (round 4): 192 x 256 output tiles — the full-tile phase schedule with a SHORT a1 quadrant: wave tile 96 x 64 = four
// fragment rows in a0 + two in a1, phases (a0,b0) (a0,b1) (a1,b1) (a1,b0) of 16 / 16 / 8 / 8 MFMAs; G3 (the a1 rows of both wave
// rows) is 64 rows = ONE DMA instruction per wave, so the counted waits that leave the three youngest groups in flight are
// vmcnt(6 / 5 / 5 / 5) (2+2+2, 1+2+2, 2+1+2, 2+2+1) and the tails 3 / 1.  A 32-image shard's launches are 2.25 (first layer) or
// 1.125 (mlp) rounds of 256-row tiles and exactly 3 / 1.5 rounds of these (gemm_route).  Plain launches only (XMODE 0).
template <typename TI, typename TO, int AMODE, bool TRAIN_EPI, bool HALF = false, int XMODE = 0, int PROBE = 0, bool T192 = false>
__global__ void __launch_bounds__(512, 2)
gemm8_kernel(const GemmArgs p, const int tiles_m, const int tiles_n, const int xcd_swizzle) {
    // (always persistent: the one-tile-per-workgroup form — 1..8 % slower, profiles/README.md — was removed in round 4)
    using X8 = typename Vec<TI>::x8;
    constexpr int BM = HALF ? G8_BM / 2 : (T192 ? 192 : G8_BM), BN = G8_BN, WM = BM / 2, WN = G8_WN;
    static_assert(!T192 || (!HALF && XMODE == 0 && AMODE <= 1 && !TRAIN_EPI), "192-row tiles: plain launches, K-contiguous operands");
    constexpr int FA1 = T192 ? 2 : 4;                   // fragment rows of the a1 quadrant
    constexpr bool A_KMAJOR = (AMODE == 3 || AMODE == 4), W_KMAJOR = (AMODE == 3);
    static_assert(!HALF || !A_KMAJOR, "half tiles: K-contiguous operands");
    constexpr int FM = WM / 16, FN = WN / 16;          // 8 x 4 (HALF: 4 x 4) accumulator fragments per wave
    constexpr int KBUF = HALF ? 3 * G8_GROUP : G8_BUF;  // one K-tile in the ring
    constexpr int RING = HALF ? 3 * KBUF : G8_RING;     // HALF: 144 KiB (three K-tiles), else 128 KiB (two)
    constexpr bool DMA_PAR = XMODE >= 3;                // epilogue parameters by LDS-DMA, double buffered
    constexpr int PAR_SZ = DMA_PAR ? (HALF ? 5120 : 7168) : G8_PAR;
    constexpr int PAR_MR = 2048, PAR_INIT = (DMA_PAR && HALF) ? 3072 : 4096, PAR_LG = HALF ? 4096 : 5120;
    constexpr int L_PAR = RING, L_RED = RING + g8_par_bytes(HALF, XMODE), L_NEXT = L_RED + g8_red_bytes(HALF, XMODE);
    static_assert(L_NEXT + 256 == g8_lds_bytes(HALF, true, XMODE), "LDS layout");
    static_assert(XMODE < 3 || (AMODE == 0 && !TRAIN_EPI), "attention epilogues: contiguous A");
    constexpr bool S2 = G8_SCHED == 2 && !HALF && !T192;          // DMA schedule 2 (G8_SCHED)
    constexpr bool PRO_G2 = S2 && XMODE != 3;                     // ... with G2(1) in the tile's prologue
    int ring_base = 0;                                  // XMODE 3, full tiles: parity of the buffer K-tile 0 of this tile uses
    int par_buf = 0;                                    // XMODE 3 / 4: the parameter buffer of the tile being computed
    auto ring_of = [&](const int kt) __attribute__((always_inline)) -> int {
        if constexpr (HALF) return kt % 3;
        else if constexpr (XMODE == 3) return (kt ^ ring_base) & 1;
        else return kt & 1;
    };

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int g = blockIdx.y;
    const int nk_full = p.K / BK;
    int nk = nk_full;                                   // K-tiles of the tile being set up / computed
    int kt_base = 0;                                    // XMODE 1 on a triangular weight: first K-tile of the tile
    const long long ldw = p.ldw_bytes ? p.ldw_bytes : (long long)p.K * 2;

    // ---- this workgroup's tile list: L, L + L_step, ... < L_end (indices into the tile_m-major tile order) ----
    const int ntiles = tiles_m * tiles_n, nwg = gridDim.x;
    int L, L_end, L_step;
    int* queue = nullptr;                               // per-XCD queue head (tiles beyond the first round), or static
    int queue_base = 0;
    if (ntiles <= nwg) {                               // one tile per workgroup
        L = xcd_swizzle ? xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x;
        L_end = L + 1; L_step = 1;
    } else if (xcd_swizzle && (nwg & 7) == 0) {        // XCD x = bid % 8 owns a contiguous run of the tile list
        const int x = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = x * q + (x < r ? x : r);
        L = start + (blockIdx.x >> 3); L_end = start + q + (x < r ? 1 : 0); L_step = nwg >> 3;
        if (p.tile_counters) { queue = p.tile_counters + g * 8 + x; queue_base = start + L_step; }
#if TP_G8_STAGGER_NS
        // A/B build flag (round 6): the workgroups of an XCD start TP_G8_STAGGER_NS apart, so that their epilogues' store bursts (4 MiB
        // per XCD and round, at the write roof when they coincide) meet other workgroups' K loops instead of each other
        {
            const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(blockIdx.x >> 3) * TP_G8_STAGGER_NS / 10;
            while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(4);
        }
#endif
    } else {
        L = blockIdx.x; L_end = ntiles; L_step = nwg;
    }

    // ---- per-tile state ------------------------------------------------------------------------------------
    int m0, n0, tile_n;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
    long long a_tile_off_cur = 0;                      // AMODE 2: the descriptor of a K-part is rebuilt from its base
    const char* tt_w_base = nullptr;                   // AMODE 3: W base of this tile's columns (group / part resolved)
    int tt_n_loc_bytes = 0;
    int voff_a[2][2], voff_w[2][2];                    // [sub][q] per-lane DMA source offsets
    const int kslot = (lane & 7) ^ (lane >> 3);

    // A is addressed relative to the tile's first row, W relative to the tile's first output column, so the
    // 32-bit per-lane offsets stay tiny whatever the batch.  A group holds 128 rows; wave w fills rows
    // 16w..16w+15 with two 1-KiB instructions (q = 0, 1): lane -> group row rho = 16w + 8q + lane/8, 16-B slot'
    // = lane%8 holding logical slot (lane%8)^(lane/8).
    //   A groups: rho -> tile row  (rho / 64) * 128 + sub * 64 + rho % 64     (sub = 0: G0, 1: G3)
    //   W groups: rho -> tile col  (rho / 32) *  64 + sub * 32 + rho % 32     (sub = 0: G1, 1: G2)
    auto a_row_off = [&](int row) __attribute__((always_inline)) -> long long {
        if constexpr (AMODE != 0) {
            if constexpr (AMODE == 1 || AMODE == 2)
                if (p.a_region_s > 0) row = region_major_to_raster(row, p.a_region_g, p.a_region_s);
            const int b = row / p.rows_per_batch;
            const int t = row - b * p.rows_per_batch;
            return (long long)b * p.a_batch_stride_bytes + (long long)t * p.lda_bytes;
        } else {
            return (long long)row * p.lda_bytes;
        }
    };
    auto setup_tile = [&](const int Lt) __attribute__((always_inline)) {
        int tm = Lt / tiles_n;
        tile_n = Lt - tm * tiles_n;
        if (xcd_swizzle == 2 && tiles_n == 8) {
            // A/B (TP_TUNE_XCD_SWIZZLE = 2): blocks of 8 row-panels x 8 N-tiles walked as two halves of 8 x 4, so that the 32
            // tiles an XCD has resident share HALF of W (4.2 MB ~ its L2) instead of all of it — W re-reads halve, A panels
            // are fetched twice.  Total fabric traffic is the same to first order; measured: see profiles/.
            const int blk = Lt >> 6, i = Lt & 63, half = i >> 5, j = i & 31;
            if ((blk << 3) + 8 <= tiles_m) { tm = (blk << 3) + (j >> 2); tile_n = (half << 2) + (j & 3); }
        } else if (xcd_swizzle == 2 && tiles_n > 8) {
            // round 5: launches with MORE than 8 N-tiles (mlp[0], mlp[2]: 16; D = 5120: 20).  In row-panel-major order the 32 tiles an
            // XCD has resident are 2 row panels x 16 N-tiles: per K-tile they pull 2 A slabs + 16 W slabs = 18 x 32 KiB through the
            // fabric, and ALL of W once per round (PMC: mlp[2] reads 2.73 GB per launch for 0.34 GB of operands).  Walk blocks of 4 row
            // panels column group by column group (groups of 8 N-tiles, the last one narrower): 4 + 8 = 12 slabs per K-tile, -33 %.
            const int per_blk = tiles_n << 2, blk = Lt / per_blk, i = Lt - blk * per_blk;
            if ((blk << 2) + 4 <= tiles_m) {
                const int full = tiles_n >> 3, g = i >> 5;              // (a full group = 4 x 8 = 32 tiles)
                const int c0 = (g < full ? g : full) << 3, w = g < full ? 8 : tiles_n - c0, j = i - (g < full ? g << 5 : full << 5);
                const int r = j / w;
                tm = (blk << 2) + r; tile_n = c0 + (j - r * w);
            }
        }
        m0 = p.m_begin + tm * BM; n0 = tile_n * BN;
        if constexpr (XMODE == 1) {
            // statistics-only launch on an UPPER-TRIANGULAR weight (GemmArgs::tri, tp_pack_qr.hip): W[n][k] = 0 for k < n0 —
            // this tile's K loop starts at K-tile n0 / 64 (16 / 12 / 8 / 4 K-tiles for the four column tiles of E = 1024)
            kt_base = p.tri ? n0 / BK : 0;
            nk = nk_full - kt_base;
        }
        // K-major operands: per-lane source offsets of DMA instruction idx = 2 wave + q of a group — k rows 4 idx ..
        // 4 idx + 3, chunk slot lane / 8 holds column block (lane / 8) ^ (wave & 1), row (lane % 8) / 2, half lane % 2
        if constexpr (W_KMAJOR) {
            tt_w_base = p.W + g * p.w_gs;
            int n_loc = n0;
            if (p.W_parts[0]) {                         // N split over four source tensors (hidden states, no torch.cat)
                const int part = n0 / p.n_part;
                n_loc = n0 - part * p.n_part;
                tt_w_base = part == 0 ? p.W_parts[0] : part == 1 ? p.W_parts[1] : part == 2 ? p.W_parts[2] : p.W_parts[3];
            }
            tt_w_base += (long long)n_loc * 2;
            tt_n_loc_bytes = n_loc * 2;
        } else {
            rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + g * p.w_gs + (long long)n0 * ldw), 0, 0x7fffffff,
                                                       0x00020000);
        }
        long long a_tile_off = 0;
        if constexpr (!A_KMAJOR) {
            a_tile_off = a_row_off(m0);
            if constexpr (AMODE == 1 || AMODE == 2)     // region-major rows scatter inside their image: address from its start
                if (p.a_region_s > 0) a_tile_off = (long long)(m0 / p.rows_per_batch) * p.a_batch_stride_bytes;
            rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + g * p.a_gs + a_tile_off), 0, 0x7fffffff, 0x00020000);
            a_tile_off_cur = a_tile_off;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int krow = 4 * (2 * wave + q) + ((lane & 7) >> 1);        // K-major forms
                const int cb = (lane >> 3) ^ (wave & 1);
                const int rho = 16 * wave + 8 * q + (lane >> 3);                // K-contiguous forms
                if constexpr (A_KMAJOR) {
                    const int acol = (cb >> 2) * 128 + sub * 64 + (cb & 3) * 16 + (lane & 1) * 8;
                    voff_a[sub][q] = (int)(krow * p.lda_bytes) + acol * 2;
                } else {
                    int row = HALF ? m0 + rho : m0 + (rho >> 6) * 128 + sub * 64 + (rho & 63);
                    if constexpr (T192) {               // wave rows are 96 apart; G3 holds 2 x 32 rows, wave w fills 8 w .. 8 w + 7 (q = 0)
                        const int r3 = 8 * wave + (lane >> 3);
                        row = sub == 0 ? m0 + (rho >> 6) * 96 + (rho & 63) : m0 + (r3 >> 5) * 96 + 64 + (r3 & 31);
                    }
                    row = row < p.M ? row : p.M - 1;
                    // (contiguous rows: a 24-bit multiply on purpose — row strides < 8 MiB, checked on the host.  The plain
                    // form compiles to v_mad_u64_u32 with a don't-care high addend, for which hipcc picked the register a
                    // parameter prefetch was still in flight to — and answered with s_waitcnt vmcnt(0) in front of every
                    // tile's DMA prologue.)
                    if constexpr (AMODE == 0) voff_a[sub][q] = __mul24(row - m0, (int)p.lda_bytes) + kslot * 16;
                    else voff_a[sub][q] = (int)(a_row_off(row) - a_tile_off) + kslot * 16;
                }
                if constexpr (W_KMAJOR) {
                    const int wcol = (cb >> 1) * 64 + sub * 32 + (cb & 1) * 16 + (lane & 1) * 8;
                    voff_w[sub][q] = (int)(krow * ldw) + wcol * 2;
                } else {
                    const int col = (rho >> 5) * 64 + sub * 32 + (rho & 31);
                    voff_w[sub][q] = __mul24(col, (int)ldw) + kslot * 16;
                }
            }
    };

    // issue DMA group G<grp> of K-tile `kt` (of the tile set up last) into the buffer of that K-tile's parity
    auto issue = [&](auto grp_, int kt) __attribute__((always_inline)) {
        constexpr int grp = decltype(grp_)::value;
        constexpr bool is_a = (grp == 0 || grp == 3);
        constexpr int sub = (grp == 2 || grp == 3) ? 1 : 0;
        char* dst = smem + ring_of(kt) * KBUF + grp * G8_GROUP + wave * 2048;
        if constexpr (is_a ? A_KMAJOR : W_KMAJOR) {
            // K-tile kt = contraction rows r0 .. r0 + 63 of this group's range; the descriptor starts at the tile's
            // first column of row r0 and ends after the last valid row (soffset is not range-checked, so K advances
            // through the base address)
            const int r0 = kt * BK;                                         // row inside the group's range
            long long rows_left = (long long)p.tt_rows - (long long)g * p.K - r0;
            rows_left = rows_left < 0 ? 0 : (rows_left > BK ? BK : rows_left);
            const char* base; long long ld, nrec;
            if constexpr (is_a) {
                ld = p.lda_bytes;
                base = p.A + g * p.a_gs + (long long)r0 * ld + (long long)m0 * 2;
                nrec = rows_left * ld - (long long)m0 * 2;
            } else {
                ld = ldw;
                const unsigned ktg = (unsigned)(g * nk + kt);               // K-tile index over all groups
                const unsigned bidx = __umulhi(ktg, (unsigned)p.tt_bmagic); // batch = ktg / tiles_per_batch (host magic)
                const unsigned tin = ktg - bidx * (unsigned)p.tt_tpb;
                base = tt_w_base + (long long)bidx * p.w_batch_stride_bytes + (long long)tin * BK * ld;
                nrec = rows_left * ld - tt_n_loc_bytes;
            }
            nrec = nrec < 0 ? 0 : nrec;
            const unsigned long long addr = (unsigned long long)base;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
            const int nr = __builtin_amdgcn_readfirstlane((int)(unsigned)nrec);
            const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nr, 0x00020000);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + q * 1024), 16,
                                                         is_a ? voff_a[sub][q] : voff_w[sub][q], 0, 0, 0);
            return;
        }
        const int soff = (XMODE == 1 ? kt_base + kt : kt) * ROW_BYTES;      // (XMODE 1: kt_base = the triangular skip, 0 otherwise)
        if constexpr (is_a && AMODE == 2) {
            // K-tile kt lives in source kt / tpp: pick that source's base with scalar selects and rebuild the
            // descriptor (words 2, 3 are constants) — four resident descriptors cost 12 more SGPRs, which pushed
            // hipcc into VGPR-held descriptors and waterfall loops around every DMA instruction
            const int ktg = kt;
            const int tpp = p.k_part / BK, part = ktg / tpp, so = (ktg - part * tpp) * ROW_BYTES;   // wave-uniform
            const char* b = part == 0 ? p.A_parts[0] : part == 1 ? p.A_parts[1] : part == 2 ? p.A_parts[2] : p.A_parts[3];
            const unsigned long long addr = (unsigned long long)(b + a_tile_off_cur);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
            const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + q * 1024), 16, voff_a[sub][q], so, 0, 0);
        } else if constexpr (T192 && grp == 3) {        // the short a1 group: one 1-KiB instruction per wave (rows 8 w .. 8 w + 7)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(smem + ring_of(kt) * KBUF + 3 * G8_GROUP + wave * 1024), 16,
                                                     voff_a[1][0], soff, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr (is_a)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(dst + q * 1024), 16, voff_a[sub][q], soff, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(dst + q * 1024), 16, voff_w[sub][q], soff, 0, 0);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // K-tile 0 complete + G0, G1 of K-tile 1: what the first phases of a tile consume
    auto issue_prologue = [&]() __attribute__((always_inline)) {
        if constexpr (HALF) {                              // (nk >= 2, checked on the host)
            issue(I0{}, 0); issue(I1{}, 0); issue(I2{}, 0); issue(I0{}, 1); issue(I1{}, 1);
        } else {
            issue(I0{}, 0); issue(I1{}, 0); issue(I2{}, 0); issue(I3{}, 0);
            if (nk >= 2) { issue(I0{}, 1); issue(I1{}, 1); if constexpr (PRO_G2) issue(I2{}, 1); }
        }
    };

    // XMODE 3: the tile's queries -> the idle part of K-tile nk-2's buffer as [region][256 columns] fp16 (512 B per region):
    // DMA instruction q of wave w moves regions PART * 32 + 4 w + 2 q + {0, 1} (1 KiB), lane -> region + lane / 32, 16-B slot
    // lane % 32.  (M / 4 is even — checked by gemm_launch — so a pair is inside Q or past its end as a whole.)
    auto issue_q = [&](auto PART_) __attribute__((always_inline)) {
        constexpr int PART = decltype(PART_)::value;
        if constexpr (XMODE == 3) {
            const long long row0 = m0 >> 2;
            const unsigned long long addr = (unsigned long long)(p.attn_q + row0 * p.attn_ldq_bytes + (long long)n0 * 2);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
            const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
            char* dst = smem + ring_of(nk - 2) * KBUF + (HALF ? 0 : (2 + PART) * G8_GROUP) + wave * 2048;
            // the per-lane part of the address is recomputed per tile (asm: not hoisted out of the tile loop — the K loop has
            // no register to spare); the region pair travels in the scalar offset, clamped to the last pair of Q
            int l = lane;
            asm volatile("" : "+v"(l));
            const int voff = (l >> 5) * (int)p.attn_ldq_bytes + (l & 31) * 16;
            const int last_pair = (p.M >> 2) - 2 - (int)row0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int pair = PART * 32 + wave * 4 + q * 2;
                pair = pair < last_pair ? pair : last_pair;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + q * 1024), 16, voff,
                                                         pair * (int)p.attn_ldq_bytes, 0, 0);
            }
        }
    };

    // ---- fragment read offsets (swizzled; fragment rows are 16-aligned inside a group, so row & 7 == lane & 7)
    const int slot0 = (((lane >> 4)) ^ (lane & 7)) << 4, slot1 = (((4 + (lane >> 4))) ^ (lane & 7)) << 4;
    const int rd_a = (wm * 64 + (lane & 15)) * ROW_BYTES;     // + i * 2048, i = 0..3
    const int rd_a3 = T192 ? (wm * 32 + (lane & 15)) * ROW_BYTES : rd_a;   // group G3 of a 192-row tile: 32 rows per wave row
    const int rd_w = (wn * 32 + (lane & 15)) * ROW_BYTES;     // + j * 2048, j = 0..1

    f32x4 acc[FM][FN];
    X8 fa[4][2];            // A fragments of the current M-quadrant  [i][k-half]
    X8 fb[2][2][2];         // W fragments                            [b][j][k-half]

    // K-major fragment reads.  Column block cb (16 columns) of a group, 32-k step kh -> two transposing reads: lane
    // group g takes k-quads 2g and 2g + 1 of the step (DMA blocks 8 kh + 2g, + 1), i.e. k = 8g .. 8g + 7 — the very k
    // a lane holds in the K-contiguous layout, so a K-major A can meet a K-contiguous W in one MFMA.  Per-lane base =
    // that DMA block + this wave's first column block + the slot XOR (+/- 128 B for odd g, by the block's parity).
    // The reads are inline asm: hipcc guards the ds_read_tr builtin with s_waitcnt vmcnt(0) whenever an LDS-DMA is in
    // flight (it cannot tell that the ring regions differ), which would drain the queue every phase.  Nothing touches
    // the destination registers before the phase's own s_waitcnt lgkmcnt(0).
    const int tt_g = lane >> 4, tt_c = (lane & 15) * 8;
    const int tt_a_even = tt_g * 2048 + tt_c + (tt_g & 1) * 128 + wm * 512;
    const int tt_a_odd = tt_g * 2048 + tt_c - (tt_g & 1) * 128 + wm * 512;
    const int tt_w_even = tt_g * 2048 + tt_c + (tt_g & 1) * 128 + wn * 256;
    const int tt_w_odd = tt_g * 2048 + tt_c - (tt_g & 1) * 128 + wn * 256;
    auto tt_read = [&](const unsigned addr, auto OFF_) __attribute__((always_inline)) -> X8 {
        constexpr int OFF = decltype(OFF_)::value;
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 1024));
        return __builtin_bit_cast(X8, (i32x4)__builtin_shufflevector(lo, hi, 0, 1, 2, 3));
    };
    // A fragments i = 0..3 of group GRP into fa[i][kh], W fragments j = 0, 1 of group GRP into fb[B][j][kh]
    auto tt_read_a = [&](const unsigned ring, auto GRP_) __attribute__((always_inline)) {
        constexpr int GRP = decltype(GRP_)::value;
        const unsigned ae = ring + tt_a_even, ao = ring + tt_a_odd;
        fa[0][0] = tt_read(ae, std::integral_constant<int, GRP * G8_GROUP + 0 * 128 + 0 * 8192>{});
        fa[0][1] = tt_read(ae, std::integral_constant<int, GRP * G8_GROUP + 0 * 128 + 1 * 8192>{});
        fa[1][0] = tt_read(ao, std::integral_constant<int, GRP * G8_GROUP + 1 * 128 + 0 * 8192>{});
        fa[1][1] = tt_read(ao, std::integral_constant<int, GRP * G8_GROUP + 1 * 128 + 1 * 8192>{});
        fa[2][0] = tt_read(ae, std::integral_constant<int, GRP * G8_GROUP + 2 * 128 + 0 * 8192>{});
        fa[2][1] = tt_read(ae, std::integral_constant<int, GRP * G8_GROUP + 2 * 128 + 1 * 8192>{});
        fa[3][0] = tt_read(ao, std::integral_constant<int, GRP * G8_GROUP + 3 * 128 + 0 * 8192>{});
        fa[3][1] = tt_read(ao, std::integral_constant<int, GRP * G8_GROUP + 3 * 128 + 1 * 8192>{});
    };
    auto tt_read_w = [&](const unsigned ring, auto GRP_, auto B_) __attribute__((always_inline)) {
        constexpr int GRP = decltype(GRP_)::value, B = decltype(B_)::value;
        const unsigned we = ring + tt_w_even, wo = ring + tt_w_odd;
        fb[B][0][0] = tt_read(we, std::integral_constant<int, GRP * G8_GROUP + 0 * 128 + 0 * 8192>{});
        fb[B][0][1] = tt_read(we, std::integral_constant<int, GRP * G8_GROUP + 0 * 128 + 1 * 8192>{});
        fb[B][1][0] = tt_read(wo, std::integral_constant<int, GRP * G8_GROUP + 1 * 128 + 0 * 8192>{});
        fb[B][1][1] = tt_read(wo, std::integral_constant<int, GRP * G8_GROUP + 1 * 128 + 1 * 8192>{});
    };
    auto read_a = [&](const char* sb, const int grp, const int i, const int kh) __attribute__((always_inline)) -> X8 {
        return *(const X8*)(sb + grp * G8_GROUP + (grp == 3 ? rd_a3 : rd_a) + i * 2048 + (kh ? slot1 : slot0));
    };
    auto read_w = [&](const char* sb, const int grp, const int j, const int kh) __attribute__((always_inline)) -> X8 {
        return *(const X8*)(sb + grp * G8_GROUP + rd_w + j * 2048 + (kh ? slot1 : slot0));
    };


    // One phase.  P: 0..3;  ISSUE: whether this phase's DMA group exists;  WAIT: vmcnt to leave in flight
    // (-1: no wait).  The DMA target of phase P in tile t is fixed by the schedule in the file header.
    auto phase = [&](auto P_, auto ISSUE_, auto WAIT_, const int t) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        constexpr int ISSUE_CODE = (int)decltype(ISSUE_)::value;  // 0 nothing | 1 the phase's groups | 2 schedule 2, K-tile 0 without G2(1) in the prologue
        constexpr bool ISSUE = ISSUE_CODE != 0;
        constexpr int WAIT = decltype(WAIT_)::value;
        const char* sb = smem + ring_of(t) * KBUF;
        // -- memory segment ---------------------------------------------------------------------------
        const unsigned ring = (unsigned)(size_t)(lds_void*)smem + (unsigned)ring_of(t) * KBUF;   // LDS byte address
        // (probe builds skip a read from the tile's second K-tile on: the registers keep the first K-tile's REAL data — zeros would
        // lower the MFMAs' power draw and raise the clock, and the probe would measure that instead)
        if ((P == 0 && (!(PROBE & 1) || t == 0)) || (P == 1 && (!(PROBE & 2) || t == 0))) {   // W fragments of b0 (group 1) / b1 (group 2)
            constexpr int B = P & 1;
            if constexpr (W_KMAJOR) {
                if constexpr (P == 0) tt_read_w(ring, I1{}, I0{}); else tt_read_w(ring, I2{}, I1{});
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) { fb[B][j][0] = read_w(sb, 1 + B, j, 0); fb[B][j][1] = read_w(sb, 1 + B, j, 1); }
            }
        }
        if ((P == 0 && (!(PROBE & 4) || t == 0)) || (P == 2 && (!(PROBE & 8) || t == 0))) {   // A fragments of a0 (group 0) / a1 (group 3)
            constexpr int GA = P == 0 ? 0 : 3;
            if constexpr (A_KMAJOR) {
                if constexpr (P == 0) tt_read_a(ring, I0{}); else tt_read_a(ring, I3{});
            } else {
#pragma unroll
                for (int i = 0; i < (P == 0 ? 4 : FA1); ++i) { fa[i][0] = read_a(sb, GA, i, 0); fa[i][1] = read_a(sb, GA, i, 1); }
            }
        }
        if constexpr (ISSUE && (PROBE & 16)) {
        } else if constexpr (ISSUE && HALF) {
            if constexpr (P == 0) issue(I2{}, t + 1);
            if constexpr (P == 1) { issue(I0{}, t + 2); issue(I1{}, t + 2); }
        } else if constexpr (ISSUE && S2) {
            if constexpr (P == 0 && ISSUE_CODE == 2) issue(I2{}, t + 1);
            if constexpr (P == 1) issue(I3{}, t + 1);
            if constexpr (P == 2) issue(I0{}, t + 2);
            if constexpr (P == 3) { issue(I1{}, t + 2); issue(I2{}, t + 2); }
        } else if constexpr (ISSUE) {
            if constexpr (P == 0) issue(I2{}, t + 1);
            if constexpr (P == 1) issue(I3{}, t + 1);
            if constexpr (P == 2) issue(I0{}, t + 2);
            if constexpr (P == 3) issue(I1{}, t + 2);
        }
        if constexpr (WAIT >= 0) wait_vmcnt<WAIT>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // (the builtin, not inline asm — round 5: after an asm wait the compiler still believes the fragment reads outstanding and
        // threads its OWN s_waitcnt lgkmcnt(7) .. (0) between the segment's MFMAs, 16 per K-tile; 0xc07f = lgkmcnt(0), vmcnt / expcnt
        // untouched.  Already-satisfied waits turned out to be free — same-box A/B of the two builds +-0.5 %, profiles/
        // r05h_lib_ab_wait.json — so this is tidiness, as in tp_gemm_pair.hip, not speed.)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_sched_barrier(0);
        // -- matrix segment ---------------------------------------------------------------------------
        constexpr int a = (P >= 2) ? 1 : 0, b = (P == 1 || P == 2) ? 1 : 0;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (PROBE & 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i][0]), "v"(fa[i][1]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(fb[b][j][0]), "v"(fb[b][j][1]));
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < (a == 0 ? 4 : FA1); ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * a + i][2 * b + j] = Mma<TI>::run(fb[b][j][ks], fa[i][ks], acc[4 * a + i][2 * b + j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    // vmcnt left in flight: steady state 2*DEPTH; the tail counts shrink as fewer groups remain to be issued
    constexpr int DEPTH_ = (T192 || XMODE == 3) ? 3 : G8_DEPTH;     // (those two derive their counts for 3 groups in flight)
    using WS_ = std::integral_constant<int, 2 * DEPTH_>;                 // steady
    using WS1_ = std::integral_constant<int, T192 ? 5 : 2 * DEPTH_>;     // steady, phases 1..3 (192-row tiles: G3 is ONE instruction)
    using WA_ = std::integral_constant<int, T192 ? 3 : 2 * (DEPTH_ - 1)>;    // tile nk-2, phase 2: G2(nk-1), G3(nk-1) stay in flight
    using WB_ = std::integral_constant<int, T192 ? 1 : 2 * (DEPTH_ - 2)>;    // tile nk-2, phase 3: G3(nk-1)
    static_assert(DEPTH_ == 3 || !T192, "192-row tiles: wait counts derived for DEPTH = 3");
    using WC_ = std::integral_constant<int, DEPTH_ == 4 ? 2 : 0>;        // tile nk-1, phase 0
    using WD_ = std::integral_constant<int, DEPTH_ == 4 ? 0 : -1>;       // tile nk-1, phase 1
    using WN_ = std::integral_constant<int, -1>;

    // The K loop of one output tile.  On entry: the tile's DMA prologue has been issued, G0(0) and G1(0) have
    // landed (vmcnt) and a workgroup barrier has been passed since.
    auto k_loop = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (XMODE >= 2) {                         // accumulators start from a per-column constant (GemmArgs::acc_init)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                // (staged in LDS with the tile's other epilogue parameters)
                const f32x4 dv = *(const f32x4*)(smem + L_PAR + par_buf * PAR_SZ + PAR_INIT + (wn * WN + (lane >> 4) * 4 + j * 16) * 4);
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = dv;
            }
        }
        if (wm == 1) __builtin_amdgcn_s_barrier();         // the late half runs one barrier behind
        __builtin_amdgcn_sched_barrier(0);
        int t = 0;
        if constexpr (HALF) {
            using W2_ = std::integral_constant<int, 2>; using W0_ = std::integral_constant<int, 0>;
            for (; t < nk - 2; ++t) { phase(I0{}, T_{}, WS_{}, t); phase(I1{}, T_{}, WS_{}, t); }
            phase(I0{}, T_{}, WS_{}, t); phase(I1{}, F_{}, W2_{}, t); ++t;     // tile nk-2: G2(nk-1) is the last group issued
            if constexpr (XMODE == 3) { issue_q(I0{}); phase(I0{}, F_{}, W2_{}, t); }   // (the queries stay in flight)
            else phase(I0{}, F_{}, W0_{}, t);                                  // tile nk-1: drain
            phase(I1{}, F_{}, WN_{}, t);
            if (wm == 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        if constexpr (S2) {                                 // schedule 2 (G8_SCHED): waits 8 / 8 / - / 10, tails 8 / 8 / - / 4 and 2 / 0 / - / -
            using X0 = std::integral_constant<int, 0>; using X1 = std::integral_constant<int, 1>; using X2 = std::integral_constant<int, 2>;
            using W10_ = std::integral_constant<int, 10>; using W8_ = std::integral_constant<int, 8>; using W4_ = std::integral_constant<int, 4>;
            using W2_ = std::integral_constant<int, 2>; using W0_ = std::integral_constant<int, 0>;
            if constexpr (!PRO_G2) {                        // the K launch (nk = 16): K-tile 0 issues G2(1) itself
                phase(I0{}, X2{}, W8_{}, t); phase(I1{}, X1{}, W8_{}, t); phase(I2{}, X1{}, WN_{}, t); phase(I3{}, X1{}, W10_{}, t);
                ++t;
            }
            for (; t < nk - 2; ++t) {
                phase(I0{}, X0{}, W8_{}, t); phase(I1{}, X1{}, W8_{}, t); phase(I2{}, X1{}, WN_{}, t); phase(I3{}, X1{}, W10_{}, t);
            }
            if (nk >= 2) {                                  // K-tile nk-2: G3(nk-1) is the last group there is to fetch
                phase(I0{}, X0{}, W8_{}, t); phase(I1{}, X1{}, W8_{}, t); phase(I2{}, X0{}, WN_{}, t); phase(I3{}, X0{}, W4_{}, t);
                ++t;
            }
            if constexpr (XMODE == 3) {                     // K-tile nk-1: the queries go out behind G2 / G3(nk-1) and stay in flight
                issue_q(I0{}); phase(I0{}, X0{}, W4_{}, t);
                issue_q(I1{}); phase(I1{}, X0{}, W4_{}, t);
            } else {
                phase(I0{}, X0{}, W2_{}, t);
                phase(I1{}, X0{}, W0_{}, t);
            }
            phase(I2{}, X0{}, WN_{}, t);
            phase(I3{}, X0{}, WN_{}, t);
            if (wm == 0) __builtin_amdgcn_s_barrier();     // re-join: equal barrier counts for both halves
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        for (; t < nk - 2; ++t) {                           // steady state: every phase issues, 3 groups stay in flight
            phase(I0{}, T_{}, WS_{}, t);
            phase(I1{}, T_{}, WS1_{}, t);
            phase(I2{}, T_{}, WS1_{}, t);
            phase(I3{}, T_{}, WS1_{}, t);
        }
        if (nk >= 2) {                                      // tile nk-2: nothing beyond tile nk-1 to fetch
            phase(I0{}, T_{}, WS_{}, t);
            phase(I1{}, T_{}, WS1_{}, t);
            phase(I2{}, F_{}, WA_{}, t);
            phase(I3{}, F_{}, WB_{}, t);
            ++t;
        }
        if constexpr (XMODE == 3) {                         // tile nk-1: drain; the queries go out and stay in flight
            static_assert(DEPTH_ == 3, "tail wait counts");
            issue_q(I0{}); phase(I0{}, F_{}, std::integral_constant<int, 2>{}, t);
            issue_q(I1{}); phase(I1{}, F_{}, WN_{}, t);
        } else {
            phase(I0{}, F_{}, WC_{}, t);                    // tile nk-1: drain
            phase(I1{}, F_{}, WD_{}, t);
        }
        phase(I2{}, F_{}, WN_{}, t);
        phase(I3{}, F_{}, WN_{}, t);
        if (wm == 0) __builtin_amdgcn_s_barrier();         // re-join: equal barrier counts for both halves
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        // ---- persistent: walk the tile list ----------------------------------------------------------------
        // Epilogue parameters of a tile, one element per thread: threads 0..255 the tile's bias / colsum column,
        // threads 256..511 the (mean, rstd) of the tile's row.
        const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
        const float* __restrict__ colsum = (p.flags & TP_LINEAR_LN_FOLD) ? p.colsum + g * p.colsum_gs : nullptr;
        const float* __restrict__ stats_in = (p.flags & TP_LINEAR_LN_FOLD) ? p.stats_in + g * p.stats_in_gs : nullptr;
        // every thread fetches column / row (tid & 255): no divergent control flow around the loads, nothing
        // consumes them before publish_params() — they ride out the epilogue in 4 VGPRs
        const float* __restrict__ acc_init = (XMODE >= 2) ? p.acc_init + g * p.acc_init_gs : nullptr;
        float pf_bias = 0.f, pf_csum = 0.f, pf_init = 0.f;
        float2 pf_mr = make_float2(0.f, 1.f);
        // XMODE 3 / 4: the parameters of the tile set up last -> parameter buffer `buf`, ONE 1-KiB DMA instruction per wave:
        //   wave 0 bias | 1 colsum | 2 acc_init | 3 (mean, rstd) of rows 0..127 | 4 rows 128..255 (HALF: the logits, lanes
        //   0..31 head 0, 32..63 head 1) | 5, 6 the logits of head 0 / 1 (XMODE 4, full tiles) | the others repeat wave 0's
        auto issue_params = [&](const int buf) __attribute__((always_inline)) {
            // (one branch per source, not one DMA fed by a select over the sources: a select between pointers loaded from the
            // kernel-argument struct made hipcc keep the whole struct in scratch)
            auto dma1k = [&](const char* src, long long len, const int dst, const int voff) __attribute__((always_inline)) {
                len = len < 0 ? 0 : (len > 0x7fffffff ? 0x7fffffff : len);
                const unsigned long long addr = (unsigned long long)src;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
                const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
                const int nr = __builtin_amdgcn_readfirstlane((int)len);
                const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nr, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + L_PAR + buf * PAR_SZ + dst), 16, voff, 0, 0, 0);
            };
            const long long rows_left = (long long)p.M - m0;
            const int lin = lane * 16;
            if (wave == 1) dma1k((const char*)(colsum + n0), 1024, 1024, lin);
            else if (wave == 2) dma1k((const char*)(acc_init + n0), 1024, PAR_INIT, lin);
            else if (wave == 3) dma1k((const char*)(stats_in + (long long)m0 * 2), rows_left * 8, PAR_MR, lin);
            else if (!HALF && wave == 4) dma1k((const char*)(stats_in + (long long)(m0 + 128) * 2), (rows_left - 128) * 8, PAR_MR + 1024, lin);
            else if (!HALF && XMODE == 4 && (wave == 5 || wave == 6))
                dma1k((const char*)(p.attn_logits + (long long)(n0 / 128 + (wave - 5)) * p.M + m0), rows_left * 4, PAR_LG + (wave - 5) * 1024, lin);
            else if (HALF && XMODE == 4 && wave == 4)       // (rows past M read the neighbouring head / the workspace behind: never used)
                dma1k((const char*)(p.attn_logits + (long long)(n0 / 128) * p.M + m0), 0x7fffffff, PAR_LG,
                      (lane >> 5) * (p.M * 4) + (lane & 31) * 16);
            else dma1k((const char*)(bias + n0), 1024, 0, lin);
        };
        auto prefetch_params = [&]() __attribute__((always_inline)) {
            if constexpr (DMA_PAR) return;
            const int e = tid & 255;
            if (bias) pf_bias = bias[n0 + e];
            if (colsum) pf_csum = colsum[n0 + e];
            if constexpr (XMODE >= 2) pf_init = acc_init[n0 + e];
            if (stats_in) {
                int m = m0 + e;
                m = m < p.M ? m : p.M - 1;
                pf_mr = *(const float2*)(stats_in + (long long)m * 2);
            }
        };
        auto publish_params = [&]() __attribute__((always_inline)) {
            if constexpr (DMA_PAR) return;
            float* par = (float*)(smem + L_PAR);
            if (tid < 256) {
                par[tid] = pf_bias; par[BN + tid] = pf_csum;
                if constexpr (XMODE >= 2) par[4 * BN + tid] = pf_init;
            }
            else *(float2*)(par + 2 * BN + 2 * (tid - 256)) = pf_mr;
        };

        setup_tile(L);
        prefetch_params();
        if constexpr (DMA_PAR) issue_params(0);
        issue_prologue();
        // Tile order inside an XCD's run: the first round is static (workgroup i takes tile i); later tiles are drawn
        // from a per-XCD queue head, so a workgroup that starts late or shares its CU (a collective's kernels on
        // another stream) simply draws fewer tiles instead of making the whole launch wait for its static share.
        // The draw for the tile AFTER this one is issued here, a whole K loop before it is needed.
        while (true) {
            int drawn = 0;
            if (queue && tid == 0) drawn = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            publish_params();                               // (hipcc waits for `pf` here)
            if (queue && tid == 0) *(int*)(smem + L_NEXT) = drawn;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // DMA prologue landed, earlier stores retired
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            k_loop();
            // the ring is free now: request the next tile before finishing this one
            const int m0c = m0, n0c = n0, tile_nc = tile_n;
            int Ln = L + L_step;
            if (queue) Ln = queue_base + __builtin_amdgcn_readfirstlane(*(const int*)(smem + L_NEXT));
            const bool has_next = Ln < L_end;               // wave-uniform
            const char* lds_q = smem + ring_of(nk - 2) * KBUF + (HALF ? 0 : 2 * G8_GROUP);     // XMODE 3: where the queries are
            if (has_next) {
                setup_tile(Ln);
                if constexpr (XMODE == 3 && !HALF) ring_base ^= 1;
                prefetch_params();
                if constexpr (DMA_PAR) issue_params(par_buf ^ 1);
                issue_prologue();
            }
            if constexpr (XMODE == 3) {                     // the queries have landed: all that was issued before the next tile's
                if (has_next) wait_vmcnt<HALF ? 11 : 13>(); else wait_vmcnt<0>();     // 1 + 10 / 12 instructions has
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float2 mean_rstd[FM];
            if constexpr (XMODE < 3)                        // (the attention epilogues read theirs row by row)
#pragma unroll
            for (int i = 0; i < FM; ++i)
            {   // (ext_vector LDS reads: a struct-typed LDS load next to an in-flight LDS-DMA makes hipcc drain vmcnt)
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t v = *(const f32x2_t*)(smem + L_PAR + 2 * BN * 4 + (wm * WM + i * 16 + (lane & 15)) * 8);
                mean_rstd[i] = make_float2(v[0], v[1]);
            }
            if constexpr (XMODE == 3)
                attn_logits_epilogue<BM, BN, WM, WN, true>(acc, p, g, m0c, n0c, wm, wn, lane, tid, mean_rstd, smem + L_RED,
                                                           smem + L_PAR + par_buf * PAR_SZ, lds_q);
            else if constexpr (XMODE == 4)
                attn_sum_epilogue<BM, BN, WM, WN, true, PAR_LG>(acc, p, g, m0c, n0c, wm, wn, lane, mean_rstd,
                                                                smem + L_PAR + par_buf * PAR_SZ);
            else
            gemm_epilogue<TO, BM, BN, WM, WN, true, TRAIN_EPI, XMODE == 1>(acc, p, g, m0c, n0c, tile_nc, wm, wn, lane, tid,
                                                                           mean_rstd, smem + L_RED, smem + L_PAR);
            if (!has_next) break;
            L = Ln;
            if constexpr (DMA_PAR) par_buf ^= 1;
            block_sync_lds();                               // everyone is done with this tile's parameters
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------
static int g8_num_cus() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    return cus;
}

bool gemm_probes_built() {
#ifdef TP_BUILD_PROBES
    return true;
#else
    return false;
#endif
}

// Workgroups of a persistent launch: one per CU, a multiple of the 8 XCDs, minus the CUs the caller reserves for
// kernels of OTHER streams (TP_TUNE_RESERVE_CUS = r: r CUs per XCD stay free — the all-gather that overlaps the next
// forward needs somewhere to run; a persistent workgroup holds its CU's LDS and registers for the whole launch).
int gemm8_persistent_cus() {
    int r = tuning(TP_TUNE_RESERVE_CUS);
    r = r < 0 ? 0 : r;                                  // (past 7: probes — tools/probes/store_bound_probe.py runs the kernel on 4 CUs per XCD)
    const int per_xcd = g8_num_cus() / 8 - r;
    return (per_xcd < 1 ? 1 : per_xcd) * 8;
}

template <typename TI, typename TO, int AMODE, bool TRAIN_EPI, bool HALF = false, int XMODE = 0, int PROBE = 0, bool T192 = false>
static int launch8_cfg(const GemmArgs& a, hipStream_t stream) {
#ifdef TP_BUILD_PROBES                                   // (libtokenpacker_exp.so only: the timing probes produce garbage results)
    if constexpr (PROBE == 0 && AMODE == 0 && !TRAIN_EPI && !HALF && XMODE == 0 && std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value) {
        const int probe = tuning(TP_TUNE_PAIR_DEBUG) >> 4;
        if (probe == 1) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 1>(a, stream);
        if (probe == 3) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 3>(a, stream);
        if (probe == 15) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 15>(a, stream);
        if (probe == 16) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 16>(a, stream);
        if (probe == 31) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 31>(a, stream);
        if (probe == 64) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 64>(a, stream);
        if (probe == 79) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 79>(a, stream);
        if (probe == 80) return launch8_cfg<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, 80>(a, stream);
    }
#endif
    auto kern = gemm8_kernel<TI, TO, AMODE, TRAIN_EPI, HALF, XMODE, PROBE, T192>;
    constexpr int lds = g8_lds_bytes(HALF, true, XMODE);
    static_assert(lds <= 160 * 1024, "LDS budget of a CU");
    constexpr int TBM = HALF ? G8_BM / 2 : (T192 ? 192 : G8_BM);
    static DynLdsAttr attr;                             // (per device, a failure is not cached: tp_internal.h)
    const hipError_t attr_err = attr.ensure(reinterpret_cast<const void*>(kern), lds);
    if (attr_err != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", lds, hipGetErrorString(attr_err));
        return TP_ERR_LAUNCH;
    }
    const int m_end = a.m_end > 0 ? a.m_end : a.M;
    const int tiles_m = (m_end - a.m_begin + TBM - 1) / TBM, tiles_n = a.N / G8_BN;
    const int ntiles = tiles_m * tiles_n;
    int nwg = ntiles;
    const int cap = gemm8_persistent_cus();            // one workgroup per CU (140 KiB of LDS each, 156 KiB with half tiles), a multiple of the 8 XCDs
    if (nwg > cap && cap > 0) nwg = cap;
    dim3 grid((unsigned)nwg, (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, a, tiles_m, tiles_n, tuning(TP_TUNE_XCD_SWIZZLE));
    return check_launch("gemm8_kernel");
}

template <typename TI, typename TO>
static int launch8_types(const GemmArgs& a, hipStream_t stream) {
    constexpr bool HALF_OUT = !std::is_same<TO, float>::value;
    const bool train_epi = (a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD)) != 0;
    if (a.tt_rows > 0) {                                // K-major operands (weight gradients): fp32 partials only
        if (a.tile192) { set_error("tp gemm8: 192-row tiles do not take K-major operands"); return TP_ERR_INVALID_ARG; }
        if constexpr (std::is_same<TO, float>::value)
            return a.tt_w_kcontig ? launch8_cfg<TI, TO, 4, false>(a, stream) : launch8_cfg<TI, TO, 3, false>(a, stream);
        set_error("tp gemm8: K-major operands are supported with fp32 output only");
        return TP_ERR_INVALID_ARG;
    }
    const bool strided_a = a.rows_per_batch < a.M || a.a_region_s > 0;     // (region-major rows: the strided-A kernels)
    if (a.A_parts[0]) {                                // K split over four sources: the forward's first layer only
        if (a.half_tiles) { set_error("tp gemm8: half tiles do not take a multi-part A operand"); return TP_ERR_INVALID_ARG; }
        if constexpr (std::is_same<TO, f16_t>::value)
            return train_epi ? launch8_cfg<TI, TO, 2, true>(a, stream) : launch8_cfg<TI, TO, 2, false>(a, stream);
        set_error("tp gemm8: a multi-part A operand is supported for fp16 output only");
        return TP_ERR_INVALID_ARG;
    }
    const bool half = a.half_tiles != 0;
    if (a.tile192) {                                    // 192 x 256 tiles: plain launches (gemm_route)
        if (half || train_epi || (a.flags & TP_LINEAR_NO_STORE) || a.acc_init || a.attn_mode || a.m_begin != 0 || a.m_end != 0) {
            set_error("tp gemm8: 192-row tiles serve plain launches over all rows only");
            return TP_ERR_INVALID_ARG;
        }
        return strided_a ? launch8_cfg<TI, TO, 1, false, false, 0, 0, true>(a, stream)
                         : launch8_cfg<TI, TO, 0, false, false, 0, 0, true>(a, stream);
    }
    if ((a.flags & TP_LINEAR_NO_STORE) || a.acc_init) {       // the two GEMMs of the fused LayerNorm chain
        if constexpr (std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value) {
            if (strided_a || train_epi || ((a.flags & TP_LINEAR_NO_STORE) && a.acc_init) || (half && a.K < 2 * BK)) {
                set_error("tp gemm8: NO_STORE / acc_init take a contiguous A, no training epilogue, and not both at once");
                return TP_ERR_INVALID_ARG;
            }
            if (a.attn_mode == 1)
                return half ? launch8_cfg<TI, TO, 0, false, true, 3>(a, stream) : launch8_cfg<TI, TO, 0, false, false, 3>(a, stream);
            if (a.attn_mode == 2)
                return half ? launch8_cfg<TI, TO, 0, false, true, 4>(a, stream) : launch8_cfg<TI, TO, 0, false, false, 4>(a, stream);
            if (a.flags & TP_LINEAR_NO_STORE)
                return half ? launch8_cfg<TI, TO, 0, false, true, 1>(a, stream) : launch8_cfg<TI, TO, 0, false, false, 1>(a, stream);
            return half ? launch8_cfg<TI, TO, 0, false, true, 2>(a, stream) : launch8_cfg<TI, TO, 0, false, false, 2>(a, stream);
        }
        set_error("tp gemm8: NO_STORE / acc_init are built for fp16 operands and output");
        return TP_ERR_INVALID_ARG;
    }
    if (half && a.K < 2 * BK) {
        set_error("tp gemm8: half tiles need K >= 128");
        return TP_ERR_INVALID_ARG;
    }
    if (train_epi) {
        if constexpr (HALF_OUT) {
            if (half) return strided_a ? launch8_cfg<TI, TO, 1, true, true>(a, stream) : launch8_cfg<TI, TO, 0, true, true>(a, stream);
            return strided_a ? launch8_cfg<TI, TO, 1, true>(a, stream) : launch8_cfg<TI, TO, 0, true>(a, stream);
        }
        set_error("tp gemm8: training epilogues need a 16-bit output");
        return TP_ERR_INVALID_ARG;
    }
    if (half) return strided_a ? launch8_cfg<TI, TO, 1, false, true>(a, stream) : launch8_cfg<TI, TO, 0, false, true>(a, stream);
    return strided_a ? launch8_cfg<TI, TO, 1, false>(a, stream) : launch8_cfg<TI, TO, 0, false>(a, stream);
}

// Preconditions (checked by gemm_launch): N % 256 == 0, K % 64 == 0, (long long)N_tile_rows * K * 2 < 2^31.
int gemm8_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream) {
    if (a.a_k_dup) { set_error("tp gemm8: a_k_dup is served by the pair kernel and the 128-tile kernel"); return TP_ERR_INVALID_ARG; }
    if (a.tt_rows == 0 && (a.lda_bytes >= (1 << 23) || (a.ldw_bytes ? a.ldw_bytes : (long long)a.K * 2) >= (1 << 23))) {
        set_error("tp gemm8: row strides must stay below 8 MiB (24-bit offset arithmetic)");
        return TP_ERR_INVALID_ARG;
    }
    if (in_dtype == TP_BF16) {
        if (out_dtype == TP_BF16) return launch8_types<bf16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch8_types<bf16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch8_types<bf16_t, float>(a, stream);
    } else if (in_dtype == TP_F16) {
        if (out_dtype == TP_BF16) return launch8_types<f16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch8_types<f16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch8_types<f16_t, float>(a, stream);
    }
    set_error("tp gemm8: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
    return TP_ERR_INVALID_ARG;
}

}  // namespace tp
