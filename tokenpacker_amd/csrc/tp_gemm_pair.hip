// tp_gemm_pair.hip — the 256x128x64 "pair" MFMA kernel: TWO co-resident 4-wave workgroups per CU, each with its own tile
// stream, so that one workgroup's epilogue (LayerNorm-fold, GELU, attention sums, 64 issue-bound 16-byte stores per wave)
// runs UNDER the other workgroup's MFMAs instead of with the matrix pipe idle.
//
//   C[M,N] = epilogue( A[M,K] · W[N,K]^T ),  N % 128 == 0, K % 64 == 0, K >= 192   (reference builder.py:112,113,120,126-136)
//
// Why a second layout next to tp_gemm8.hip (one 8-wave workgroup per CU, 256x256 tiles): there the fixed cost per output tile
// — 8.4 us plain ... 12.4 us GELU against 23 us of main loop at K = 1024 — is paid with every SIMD's matrix pipe idle, because
// all eight waves reach the epilogue together and a wave has no register left to park a packed tile in.  A CU
// admits a second workgroup if each takes half the registers' waves (4 waves x 256 VGPRs = one wave per SIMD each) and half the
// LDS (80 KiB), and two independent workgroups need no shared vmcnt, no shared barrier and no shared ring: whatever one does
// at its tile seam, the other's K loop keeps the pipe fed.
//
//   * 4 waves (2 along M x 2 along N), wave tile 128 x 64 = 128 fp32 accumulator registers — the fragment layout, MFMA order
//     and epilogue of tp_gemm8.hip (same bits: a K-tile is consumed as the four quadrant phases (a0,b0) (a0,b1) (a1,b1) (a1,b0)
//     of 16 v_mfma_f32_16x16x32 each, 12 / 4 / 8 / 0 ds_read_b128 per phase).
//   * ONE s_barrier per phase (4 waves), placed between a phase's memory segment (fragment reads, LDS-DMA issue, counted
//     vmcnt) and its matrix segment.  The two workgroups of a CU are not synchronised with each other; on every SIMD the wave of
//     one workgroup finds the pipe free while the wave of the other is in its memory segment, and they fall into alternation.
//   * LDS per workgroup: an A ring of THREE 16-KiB groups and a W ring of THREE 8-KiB groups (a group = the rows of one
//     M-quadrant a0 / a1 of both wave rows, resp. the W rows of one N-quadrant b0 / b1 of both wave columns, x 64 k = 128-byte
//     rows, whole cache lines) — 1.5 K-tiles of each operand, 72 KiB, instead of two whole K-tiles.  Group sequence per
//     operand: a0(0) a1(0) a0(1) a1(1) ..., slot = index % 3.
//   * HBM/L2 -> LDS by buffer_load_dwordx4 ... lds, 1 KiB per wave instruction ("piece": 8 rows x 128 B; an A group is 4
//     pieces per wave, a W group 2), same lane-linear image + source-side swizzle as tp_gemm8.hip.  While tile u is computed
//     the 12 pieces  b1(u+1)[2] a1(u+1)[4] | b0(u+2)[2] a0(u+2)[4]  are issued 3 / 3 / 4 / 2 in phases 1, 2, 3 of tile u and
//     phase 0 of tile u + 1:
//         WAR  a group's slot was last read one group-triple earlier: a1(u+1) replaces a0(u) (read in phase 0 of u, issued
//              from phase 1), b1(u+1) replaces b0(u) (phase 0 -> 1), b0(u+2) replaces b1(u) (phase 1 -> issued in 3),
//              a0(u+2) replaces a1(u) (phase 2 -> 3); reads retire at the lgkmcnt(0) in front of the phase's barrier, the
//              issue segment of the next phase lies behind it.
//         RAW  the counted wait that covers a group sits in front of the barrier of the phase BEFORE its first read:
//              b1(u) in phase 0 of u: vmcnt(10); a1(u) in phase 1: vmcnt(9); a0(u+1), b0(u+1) in phase 3: vmcnt(10) — every
//              piece has been in flight for at least 3 phases by then (counts: see k_loop()).
//   * persistent: 2 workgroups per CU walk per-XCD tile runs (first round static, later tiles drawn from the per-XCD queue
//     heads, like tp_gemm8.hip); when a tile's K loop ends, the next tile's 18-piece DMA prologue is issued BEFORE the epilogue
//     runs.  Epilogue parameters travel HBM/L2 -> LDS by DMA as well (no register carries them through an epilogue): bias,
//     colsum, the rows' (mean, rstd) and the V launch's logits into ONE buffer, issued right behind the barrier that opens the
//     tile's K loop (the previous tile's epilogue has finished with the buffer; they are older than every piece the loop's counted
//     waits retire, so they have landed long before the epilogue reads them); acc_init — which the K loop reads first thing —
//     into one of two buffers, issued with the prologue a tile ahead.  The epilogue's reduction scratch is W-ring slot 2,
//     which the prologue leaves empty (b0(1) is issued in phase 0 of K-tile 0 instead).
//   * the workgroups of the second half of the grid (the ones the dispatcher places second on each CU) start `stagger`
//     sleep quanta late: two workgroups that share a pipe evenly finish their tiles TOGETHER for ever (a lead is preserved,
//     never created), and then both epilogues would still coincide.  Half a tile period apart, each epilogue falls into the
//     middle of the other's K loop.  Placement is a speed heuristic only; nothing depends on it for correctness.
#include "tp_gemm_common.h"
#include <atomic>
#include <mutex>

namespace tp {

namespace {

template <int N> __device__ __forceinline__ void gp_wait_vmcnt() {
    static_assert(N == 0 || N == 4 || N == 6 || N == 9 || N == 10, "unsupported count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}

constexpr int GP_BM = 256, GP_BN = 128, GP_WM = 128, GP_WN = 64;
constexpr int GP_AGRP = 128 * ROW_BYTES;            // 16 KiB: the rows of one M-quadrant (a0 or a1) of both wave rows
constexpr int GP_WGRP = 64 * ROW_BYTES;             //  8 KiB: the W rows of one N-quadrant (b0 or b1) of both wave columns
constexpr int GP_L_W = 3 * GP_AGRP;                 // W ring behind the A ring
constexpr int GP_L_PAR = GP_L_W + 3 * GP_WGRP;      // 72 KiB of rings, then the epilogue parameters of the tile:
// bias[128] | colsum[128] | (mean, rstd)[256] — gemm_epilogue's LDS_PARAMS layout for BN = 128 — | acc_init[128] x 2 | logits[256]
constexpr int GP_PAR_MR = 2 * GP_BN * 4, GP_PAR_INIT = GP_PAR_MR + GP_BM * 8, GP_PAR_LG = GP_PAR_INIT + 2 * GP_BN * 4;
constexpr int GP_L_NEXT = GP_L_PAR + GP_PAR_LG + GP_BM * 4;
constexpr int GP_L_RED = GP_L_W + 2 * GP_WGRP;      // the epilogue's reduction scratch: W-ring slot 2 (<= 4 KiB of its 8)
constexpr int GP_LDS_BYTES = GP_L_NEXT + 16;
static_assert(GP_LDS_BYTES <= 80 * 1024, "two workgroups per CU");

}  // namespace

// AMODE: 0 = A rows contiguous (lda), 1 = rows in batches with a batch stride (the CLIP tower's [:,1:] slices), optionally in
//        region-major order (GemmArgs::a_region_s), 2 = like 1 with K split over four source tensors (tp_gemm8.hip).
// XMODE: 0 plain | 1 statistics only, optionally on a triangular weight | 2 accumulators pre-loaded from GemmArgs::acc_init |
//        3 / 4: 2 + region attention in the epilogue (the K launch / the V launch; tp_gemm_common.h).  The K launch's queries
//        are read from global memory by the epilogue (the other workgroup of the CU covers their latency).
// DBG (TP_TUNE_PAIR_DEBUG, probe builds for timing only — results are garbage): 1 no DMA in the K loop | 2 no fragment reads |
// 3 no barriers | 4 no MFMAs
template <typename TI, typename TO, int AMODE, int XMODE, int DBG = 0>
__global__ void __launch_bounds__(256, 2)
gemm_pair_kernel(const GemmArgs p, const int tiles_m, const int tiles_n, const int stagger) {
    using X8 = typename Vec<TI>::x8;
    constexpr int BM = GP_BM, BN = GP_BN, WM = GP_WM, WN = GP_WN;
    constexpr int FM = WM / 16, FN = WN / 16;           // 8 x 4 accumulator fragments per wave
    constexpr int L_PAR = GP_L_PAR, L_NEXT = GP_L_NEXT;
    int init_buf = 0;                                   // XMODE >= 2: the acc_init buffer of the tile being computed

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    if (ntiles <= nwg) {
        L = xcd_remap(blockIdx.x, nwg); L_end = L + 1; L_step = 1;
    } else if ((nwg & 7) == 0) {                        // XCD x = bid % 8 owns a contiguous run of the tile list
        const int x = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        const int start = x * q + (x < r ? x : r);
        L = start + (blockIdx.x >> 3); L_end = start + q + (x < r ? 1 : 0); L_step = nwg >> 3;
        if (p.tile_counters) { queue = p.tile_counters + g * 8 + x; queue_base = start + L_step; }
    } else {
        L = blockIdx.x; L_end = ntiles; L_step = nwg;
    }
    if (L >= L_end) return;                             // (uniform per workgroup; nothing has been issued yet)

    // the second workgroup of every CU starts half a tile period late (header)
    if (stagger > 0 && (int)blockIdx.x >= (nwg >> 1))
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);

    // ---- per-tile state ------------------------------------------------------------------------------------
    int m0, n0, tile_n;
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
    long long a_tile_off_cur = 0;                      // AMODE 2: the descriptor of a K-part is rebuilt from its base
    int voff_a[2][4], voff_w[2][2];                    // [sub][piece] per-lane DMA source offsets
    const int kslot = (lane & 7) ^ (lane >> 3);

    // A group `sub` holds 128 rows: group row rho -> tile row (rho / 64) * 128 + sub * 64 + rho % 64; wave w fills rho =
    // 32 w .. 32 w + 31 with four 1-KiB pieces (q = 0..3: 8 rows each), lane -> rho = 32 w + 8 q + lane / 8, 16-B slot' = lane % 8
    // holding logical k-slot (lane % 8) ^ (lane / 8).  W group `sub`: 64 rows, rho -> tile column (rho / 32) * 64 + sub * 32 +
    // rho % 32, wave w fills rho = 16 w .. 16 w + 15 with two pieces.
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
    auto setup_tile = [&](const int Lt) __attribute__((always_inline)) {
        const int tm = Lt / tiles_n;
        tile_n = Lt - tm * tiles_n;
        m0 = tm * BM; n0 = tile_n * BN;
        if constexpr (XMODE == 1) {                     // triangular weight: W[n][k] = 0 for k < n0 (tp_pack_qr.hip) — the K loop
            // starts at K-tile n0 / 64, but keeps at least 4 K-tiles (the ring schedule needs 3; the extra ones multiply stored zeros)
            kt_base = p.tri ? (n0 / BK < nk_full - 4 ? n0 / BK : (nk_full > 4 ? nk_full - 4 : 0)) : 0;
            nk = nk_full - kt_base;
        }
        rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + g * p.w_gs + (long long)n0 * ldw), 0, 0x7fffffff, 0x00020000);
        long long a_tile_off = a_row_off(m0);
        if constexpr (AMODE != 0)                       // region-major rows scatter inside their image: address from its start
            if (p.a_region_s > 0) a_tile_off = (long long)(m0 / p.rows_per_batch) * p.a_batch_stride_bytes;
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + g * p.a_gs + a_tile_off), 0, 0x7fffffff, 0x00020000);
        a_tile_off_cur = a_tile_off;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rho = 32 * wave + 8 * q + (lane >> 3);
                int row = m0 + (rho >> 6) * 128 + sub * 64 + (rho & 63);
                row = row < p.M ? row : p.M - 1;
                // (24-bit multiply on purpose — tp_gemm8.hip: the 64-bit form drew a vmcnt(0) from hipcc)
                if constexpr (AMODE == 0) voff_a[sub][q] = __mul24(row - m0, (int)p.lda_bytes) + kslot * 16;
                else voff_a[sub][q] = (int)(a_row_off(row) - a_tile_off) + kslot * 16;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rho = 16 * wave + 8 * q + (lane >> 3);
                const int col = (rho >> 5) * 64 + sub * 32 + (rho & 31);
                voff_w[sub][q] = __mul24(col, (int)ldw) + kslot * 16;
            }
        }
    };

    // Ring slots: group `sub` of K-tile kt has sequence index 2 kt + sub and lives in slot index % 3.  The K loop keeps the BYTE
    // OFFSETS of the slots it reads in four scalars that advance cyclically (three SALU instructions per advance) — a_p0 / a_p2:
    // the A slots read in phases 0 / 2 (a0, a1 of the K-tile being computed), w_p0 / w_p1: the W slots read in phases 0 / 1 —
    // and every piece it issues lands in a slot it has JUST finished reading (header: a1(u+1) -> a0(u)'s, b1(u+1) -> b0(u)'s,
    // b0(u+2) -> b1(u)'s, a0(u+2) -> a1(u)'s), so no write cursor exists.  A fragment read is base VGPR + slot offset (one
    // v_add per k-half and group) + immediates; with `% 3` arithmetic on the K-tile index instead the loop carried ~90 scalar /
    // vector ALU instructions per K-tile beside its 64 MFMAs, and a wave issues one instruction per ~4 cycles: the memory
    // segments were issue-bound (measured: 1.80 instead of 1.45 us per K-tile and 256 x 256 tile).
    // K advances through the DMA instructions' scalar offset: soff_k = byte offset of K-tile u + 1 (pieces of tile u + 2: + 128).
    // The A operand's own K offsets for the pieces one / two K-tiles ahead (D = 1 / 2): equal to soff_k, soff_k + 128 unless
    // GemmArgs::a_k_dup lets the leading K-tiles of A serve two K-tiles of W each (a_map; six SALU instructions per K-tile in advance()).
    int soff_k = 0, soff_a1 = 0, soff_a2 = 0, a_step = 0;            // a_step: the K-tile soff_a2 belongs to
    const int a_dup_tiles = p.a_k_dup / BK;
    auto a_map = [&](const int j) __attribute__((always_inline)) -> int { return j < 2 * a_dup_tiles ? j >> 1 : j - a_dup_tiles; };
    int a_p0 = 0, a_p2 = 0, w_p0 = 0, w_p1 = 0;
    auto next_a = [&](const int off) __attribute__((always_inline)) -> int { return off == 2 * GP_AGRP ? 0 : off + GP_AGRP; };
    auto next_w = [&](const int off) __attribute__((always_inline)) -> int { return off == 2 * GP_WGRP ? 0 : off + GP_WGRP; };
    // one 1-KiB piece of an A / W group into the slot at byte offset `slot_off`; the K-tile it belongs to lies D tiles ahead of the
    // one being computed (prologue: D = 0 / 1 for K-tiles 0 / 1 with soff_k preset)
    auto issue_a = [&](auto SUB_, auto Q_, const int slot_off, auto D_) __attribute__((always_inline)) {
        constexpr int sub = decltype(SUB_)::value, q = decltype(Q_)::value, D = decltype(D_)::value;
        char* dst = smem + slot_off + (wave * 4 + q) * 1024;
        const int soff = D == 0 ? soff_k - ROW_BYTES : (D == 1 ? soff_a1 : soff_a2);      // (D = 0: K-tile 0 of the prologue)
        if constexpr (AMODE == 2) {
            // the K-tile lives in source ktg / tpp: pick that source's base with scalar selects and rebuild the descriptor
            const int ktg = soff / ROW_BYTES;
            const int tpp = p.k_part / BK, part = ktg / tpp, so = (ktg - part * tpp) * ROW_BYTES;   // wave-uniform
            const char* b = part == 0 ? p.A_parts[0] : part == 1 ? p.A_parts[1] : part == 2 ? p.A_parts[2] : p.A_parts[3];
            const unsigned long long addr = (unsigned long long)(b + a_tile_off_cur);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
            const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, voff_a[sub][q], so, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)dst, 16, voff_a[sub][q], soff, 0, 0);
        }
    };
    auto issue_w = [&](auto SUB_, auto Q_, const int slot_off, auto D_) __attribute__((always_inline)) {
        constexpr int sub = decltype(SUB_)::value, q = decltype(Q_)::value, D = decltype(D_)::value;
        char* dst = smem + GP_L_W + slot_off + (wave * 2 + q) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)dst, 16, voff_w[sub][q], soff_k + (D - 1) * ROW_BYTES, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // K-tile 0 complete + a0 of K-tile 1 (nk >= 3, checked on the host), in the order of first use.  W slot 2 stays empty: it is
    // the epilogue's scratch while this prologue is in flight; b0(1) follows in phase 0 of K-tile 0.
    auto issue_prologue = [&]() __attribute__((always_inline)) {
        soff_k = (kt_base + 1) * ROW_BYTES;                 // (K-tile 0 = "D = 0", K-tile 1 = D = 1)
        soff_a1 = (kt_base + a_map(1)) * ROW_BYTES; soff_a2 = (kt_base + a_map(2)) * ROW_BYTES; a_step = 2;
        // sequence indices 0, 1, 2 -> slots 0, 1, 2:  a0(0) b0(0) | b1(0) a1(0) | a0(1)
        issue_a(I0{}, I0{}, 0, I0{}); issue_a(I0{}, I1{}, 0, I0{}); issue_a(I0{}, I2{}, 0, I0{}); issue_a(I0{}, I3{}, 0, I0{});
        issue_w(I0{}, I0{}, 0, I0{}); issue_w(I0{}, I1{}, 0, I0{});
        issue_w(I1{}, I0{}, GP_WGRP, I0{}); issue_w(I1{}, I1{}, GP_WGRP, I0{});
        issue_a(I1{}, I0{}, GP_AGRP, I0{}); issue_a(I1{}, I1{}, GP_AGRP, I0{}); issue_a(I1{}, I2{}, GP_AGRP, I0{}); issue_a(I1{}, I3{}, GP_AGRP, I0{});
        issue_a(I0{}, I0{}, 2 * GP_AGRP, I1{}); issue_a(I0{}, I1{}, 2 * GP_AGRP, I1{}); issue_a(I0{}, I2{}, 2 * GP_AGRP, I1{}); issue_a(I0{}, I3{}, 2 * GP_AGRP, I1{});
    };

    // ---- fragment read offsets (swizzled; fragment rows are 16-aligned inside a group, so row & 7 == lane & 7)
    const int slot0 = (((lane >> 4)) ^ (lane & 7)) << 4, slot1 = (((4 + (lane >> 4))) ^ (lane & 7)) << 4;
    // per-lane read bases (k-half 0 / 1); everything else of a fragment's address — ring slot, fragment index — is an immediate
    const char* const rd_a0 = smem + (wm * 64 + (lane & 15)) * ROW_BYTES + slot0;             // + slot * GP_AGRP + i * 2048, i = 0..3
    const char* const rd_a1 = smem + (wm * 64 + (lane & 15)) * ROW_BYTES + slot1;
    const char* const rd_w0 = smem + GP_L_W + (wn * 32 + (lane & 15)) * ROW_BYTES + slot0;    // + slot * GP_WGRP + j * 2048, j = 0..1
    const char* const rd_w1 = smem + GP_L_W + (wn * 32 + (lane & 15)) * ROW_BYTES + slot1;

    f32x4 acc[FM][FN];
    X8 fa[4][2];            // A fragments of the current M-quadrant  [i][k-half]
    X8 fb[2][2][2];         // W fragments                            [b][j][k-half]
    if constexpr (DBG == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i][0] = X8{}; fa[i][1] = X8{}; }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) { fb[b][j][0] = X8{}; fb[b][j][1] = X8{}; }
    }

    // One phase of the K-tile being computed.  P: 0..3;  ISSUE: whether this phase's DMA pieces exist;  WAIT: vmcnt to leave in
    // flight (-1: none).  FIRST: phase 0 of K-tile 0 — what the prologue left out (b0(1), into W slot 2) instead of the steady-state
    // pieces.  Slot cursors on entry to phase 0 of K-tile u: a_p0 = slot of a0(u), a_p2 = of a1(u), w_p0 = of b0(u), w_p1 = of b1(u).
    auto phase = [&](auto P_, auto ISSUE_, auto WAIT_, auto FIRST_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        constexpr bool ISSUE = decltype(ISSUE_)::value;
        constexpr int WAIT = decltype(WAIT_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value;
        // -- memory segment ---------------------------------------------------------------------------
        if constexpr ((P == 0 || P == 1) && DBG != 2) {             // W fragments of b0 / b1
            constexpr int B = P;
            const int off = B == 0 ? w_p0 : w_p1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[B][j][0] = *(const X8*)(rd_w0 + off + j * 2048);
                fb[B][j][1] = *(const X8*)(rd_w1 + off + j * 2048);
            }
        }
        if constexpr ((P == 0 || P == 2) && DBG != 2) {             // A fragments of a0 / a1
            const int off = P == 0 ? a_p0 : a_p2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][0] = *(const X8*)(rd_a0 + off + i * 2048);
                fa[i][1] = *(const X8*)(rd_a1 + off + i * 2048);
            }
        }
        if constexpr (FIRST && DBG != 1) { issue_w(I0{}, I0{}, 2 * GP_WGRP, I1{}); issue_w(I0{}, I1{}, 2 * GP_WGRP, I1{}); }
        if constexpr (ISSUE && DBG != 1) {
            // P0: a0(u+1) pieces 2, 3 -> the slot of a1(u-1) = the one AFTER a0(u)'s in the cycle... see the cursor notes below
            if constexpr (P == 0) { const int d = next_a(a_p2); issue_a(I0{}, I2{}, d, I1{}); issue_a(I0{}, I3{}, d, I1{}); }
            if constexpr (P == 1) { issue_w(I1{}, I0{}, w_p0, I1{}); issue_w(I1{}, I1{}, w_p0, I1{}); issue_a(I1{}, I0{}, a_p0, I1{}); }
            if constexpr (P == 2) { issue_a(I1{}, I1{}, a_p0, I1{}); issue_a(I1{}, I2{}, a_p0, I1{}); issue_a(I1{}, I3{}, a_p0, I1{}); }
            if constexpr (P == 3) { issue_w(I0{}, I0{}, w_p1, I2{}); issue_w(I0{}, I1{}, w_p1, I2{}); issue_a(I0{}, I0{}, a_p2, I2{}); issue_a(I0{}, I1{}, a_p2, I2{}); }
        }
        if constexpr (WAIT >= 0) gp_wait_vmcnt<WAIT>();
        __builtin_amdgcn_sched_barrier(0);
        // (the builtin, not inline asm: the compiler then KNOWS the counter is zero and adds no lgkmcnt waits of its own in
        // front of the MFMAs — six wasted issue slots per phase; 0xc07f = lgkmcnt(0), vmcnt / expcnt untouched)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if constexpr (DBG != 3) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // -- matrix segment ---------------------------------------------------------------------------
        constexpr int a = (P >= 2) ? 1 : 0, b = (P == 1 || P == 2) ? 1 : 0;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (DBG != 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * a + i][2 * b + j] = Mma<TI>::run(fb[b][j][ks], fa[i][ks], acc[4 * a + i][2 * b + j]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(fa[i][0]), "v"(fa[i][1])); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { asm volatile("" :: "v"(fb[b][j][0]), "v"(fb[b][j][1])); }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using W10_ = std::integral_constant<int, 10>; using W9_ = std::integral_constant<int, 9>;
    using W6_ = std::integral_constant<int, 6>; using W4_ = std::integral_constant<int, 4>;
    using W0_ = std::integral_constant<int, 0>; using WN_ = std::integral_constant<int, -1>;

    // The K loop of one output tile.  On entry: the tile's 18-piece prologue has landed and a workgroup barrier has been
    // passed since.  Per-wave DMA queue in steady state (u.i = piece i of the 12 issued while tile u is computed):
    //   P1(u): u.0 u.1 (b1(u+1)) u.2 | P2(u): u.3 u.4 u.5 (a1(u+1)) | P3(u): u.6 u.7 (b0(u+2)) u.8 u.9 | P0(u+1): u.10 u.11 (a0(u+2))
    //   P0(u) needs b1(u)   = (u-1).0-1  : younger (u-1).2-11                -> vmcnt(10)   [4 when nothing was issued for u+1]
    //   P1(u) needs a1(u)   = (u-1).2-5  : younger (u-1).6-11, u.0-2         -> vmcnt(9)    [0 in the last tile]
    //   P3(u) needs a0(u+1), b0(u+1) = (u-1).6-11 : younger u.0-9            -> vmcnt(10)   [6 when nothing was issued for u+2]
    auto k_loop = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (XMODE >= 2) {                         // accumulators start from a per-column constant (GemmArgs::acc_init)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const f32x4 dv = *(const f32x4*)(smem + L_PAR + GP_PAR_INIT + init_buf * (BN * 4) + (wn * WN + (lane >> 4) * 4 + j * 16) * 4);
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = dv;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // Slot cursors: sequence a0(0) a1(0) a0(1) a1(1) ... over slots 0 1 2 0 1 2 ...  =>  a0(u+1) = next(a1(u)), a1(u+1) = next(a0(u+1))
        // = the slot of a0(u) (three slots: next(next(next(x))) = x), likewise for W.  Where the pieces of phase P go (header):
        //   P1, P2: a1(u+1) -> a0(u)'s slot = a_p0;  b1(u+1) -> b0(u)'s = w_p0        (read in phase 0)
        //   P3:     b0(u+2) -> b1(u)'s = w_p1 (read in phase 1);  a0(u+2) -> a1(u)'s = a_p2 (read in phase 2), pieces 0, 1
        //   P0 of u+1: a0(u+2) pieces 2, 3 -> a1(u)'s slot, which after advance() is next(a_p2)  [a_p2 now = a1(u+1) = a0(u)'s old slot;
        //           next of it = a1(u)'s old slot: the cycle a0(u) -> a1(u) -> a0(u+1) -> a0(u)]
        a_p0 = 0; a_p2 = GP_AGRP; w_p0 = 0; w_p1 = GP_WGRP;
        auto advance = [&]() __attribute__((always_inline)) {   // cursors of K-tile u -> K-tile u + 1
            const int a0n = next_a(a_p2), w0n = next_w(w_p1);
            a_p2 = a_p0; w_p1 = w_p0;                       // a1(u+1) took a0(u)'s slot, b1(u+1) b0(u)'s
            a_p0 = a0n; w_p0 = w0n;
            soff_k += ROW_BYTES;
            soff_a1 = soff_a2;
            ++a_step;
            soff_a2 = (kt_base + a_map(a_step)) * ROW_BYTES;
        };
        // K-tile kinds: steady (every phase issues), last but one (nothing beyond K-tile nk - 1 to fetch), last (drain)
        // K-tile 0: a0(1) came whole with the prologue; b0(1) goes out in its phase 0
        phase(I0{}, F_{}, WN_{}, T_{}); phase(I1{}, T_{}, W9_{}, F_{}); phase(I2{}, T_{}, WN_{}, F_{}); phase(I3{}, T_{}, W10_{}, F_{});
        advance();
        for (int u = 1; u < nk - 2; ++u) {                  // steady state
            phase(I0{}, T_{}, W10_{}, F_{}); phase(I1{}, T_{}, W9_{}, F_{}); phase(I2{}, T_{}, WN_{}, F_{}); phase(I3{}, T_{}, W10_{}, F_{});
            advance();
        }
        phase(I0{}, T_{}, W10_{}, F_{}); phase(I1{}, T_{}, W9_{}, F_{}); phase(I2{}, T_{}, WN_{}, F_{}); phase(I3{}, F_{}, W6_{}, F_{});
        advance();
        phase(I0{}, F_{}, W4_{}, F_{}); phase(I1{}, F_{}, W0_{}, F_{}); phase(I2{}, F_{}, WN_{}, F_{}); phase(I3{}, F_{}, WN_{}, F_{});
    };

    // ---- persistent: walk the tile list ----------------------------------------------------------------
    const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
    const float* __restrict__ colsum = (p.flags & TP_LINEAR_LN_FOLD) ? p.colsum + g * p.colsum_gs : nullptr;
    const float* __restrict__ stats_in = (p.flags & TP_LINEAR_LN_FOLD) ? p.stats_in + g * p.stats_in_gs : nullptr;
    const float* __restrict__ acc_init = (XMODE >= 2) ? p.acc_init + g * p.acc_init_gs : nullptr;
    // one DMA piece of parameters: `len` bytes at src -> the parameter area at dst (lane l moves bytes 16 l .. 16 l + 15; HALF:
    // a 512-byte array — lanes 0..31 only, the others neither load nor write; bytes past `len` arrive as zeros)
    auto dma_par = [&](const char* src, long long len, const int dst, auto HALF_) __attribute__((always_inline)) {
        constexpr bool HALF = decltype(HALF_)::value;
        len = len < 0 ? 0 : (len > 0x7fffffff ? 0x7fffffff : len);
        const unsigned long long addr = (unsigned long long)src;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
        const int nr = __builtin_amdgcn_readfirstlane((int)len);
        const auto r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nr, 0x00020000);
        if (!HALF || lane < 32)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + L_PAR + dst), 16, lane * 16, 0, 0, 0);
    };
    // acc_init of the tile set up last -> init buffer `buf` (with the tile's prologue, a tile ahead)
    auto issue_init = [&](const int buf) __attribute__((always_inline)) {
        if constexpr (XMODE >= 2)
            if (wave == 1) dma_par((const char*)(acc_init + n0), BN * 4, GP_PAR_INIT + buf * (BN * 4), T_{});
    };
    // the other parameters of tile (m0t, n0t) -> the single buffer the epilogues read (behind the barrier that opens the K loop)
    auto issue_params = [&](const int m0t, const int n0t) __attribute__((always_inline)) {
        const long long rows_left = (long long)p.M - m0t;
        if (wave == 0) {
            if (bias) dma_par((const char*)(bias + n0t), BN * 4, 0, T_{});
            if constexpr (XMODE == 4)
                dma_par((const char*)(p.attn_logits + (long long)(n0t / 128) * p.M + m0t), rows_left * 4, GP_PAR_LG, F_{});
        } else if (wave == 1) {
            if (colsum) dma_par((const char*)(colsum + n0t), BN * 4, BN * 4, T_{});
        } else if (stats_in) {
            const int half = wave - 2;                  // rows 0..127 / 128..255 of the tile
            dma_par((const char*)(stats_in + (long long)(m0t + half * 128) * 2), (rows_left - half * 128) * 8, GP_PAR_MR + half * 1024, F_{});
        }
    };
    if (!bias && tid < BN) ((float*)(smem + L_PAR))[tid] = 0.f;      // (the epilogues add the staged bias unconditionally)

    setup_tile(L);
    issue_init(0);
    issue_prologue();
    // Tile order inside an XCD's run: the first round is static, later tiles are drawn from the per-XCD queue head.  The draw that
    // decides the tile after next is issued BEFORE an epilogue and consumed after it (the atomic's round trip is ~1 us under load).
    int drawn = 0;
    if (queue && tid == 0) drawn = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        if (queue && tid == 0) *(int*)(smem + L_NEXT) = drawn;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // DMA prologue + acc_init landed, earlier stores retired
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_params(m0, n0);
        k_loop();
        // (every wave has passed the last phase's barrier: all fragment reads have retired and the rings are free)
        const int m0c = m0, n0c = n0, tile_nc = tile_n;
        int Ln = L + L_step;
        if (queue) Ln = queue_base + __builtin_amdgcn_readfirstlane(*(const int*)(smem + L_NEXT));
        const bool has_next = Ln < L_end;               // wave-uniform
        float2 mean_rstd[FM];                           // (unused: every epilogue reads its rows' (mean, rstd) from LDS)
#pragma unroll
        for (int i = 0; i < FM; ++i) mean_rstd[i] = make_float2(0.f, 1.f);
        if (has_next) {
            setup_tile(Ln);
            issue_init(init_buf ^ 1);
            issue_prologue();
            if (queue && tid == 0) drawn = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the epilogue's per-lane offsets are recomputed per tile from a laundered thread id: hoisted out of the tile loop they
        // would ride through the K loop in registers it does not have — hipcc spilled up to 23 of them)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63;
        if constexpr (XMODE == 3)
            attn_logits_epilogue<BM, BN, WM, WN, true, false>(acc, p, g, m0c, n0c, wm, wn, lane_e, tid_e, mean_rstd, smem + GP_L_RED,
                                                              smem + L_PAR, nullptr);
        else if constexpr (XMODE == 4)
            attn_sum_epilogue<BM, BN, WM, WN, true, GP_PAR_LG>(acc, p, g, m0c, n0c, wm, wn, lane_e, mean_rstd, smem + L_PAR);
        else
            gemm_epilogue<TO, BM, BN, WM, WN, true, false, XMODE == 1, true>(acc, p, g, m0c, n0c, tile_nc, wm, wn, lane_e, tid_e, mean_rstd,
                                                                             smem + GP_L_RED, smem + L_PAR);
        if (!has_next) break;
        L = Ln;
        init_buf ^= 1;
        block_sync_lds();                               // everyone is done with this tile's parameters and scratch
    }
}

// ---- host side ------------------------------------------------------------------------------------------
// Whether gemm_launch may send `a` to the pair kernel (shape / feature coverage; the routing POLICY is gemm_launch's).
bool gemm_pair_supports(int in_dtype, int out_dtype, const GemmArgs& a) {
    if (a.tt_rows > 0 || a.half_tiles || a.m_begin != 0 || a.m_end != 0 || a.stats_parts || a.parts_k_groups) return false;
    if (a.flags & (TP_LINEAR_SAVE_PRE | TP_LINEAR_GELU_BWD)) return false;
    if (a.N % GP_BN != 0 || a.K % BK != 0 || a.K / BK < 3) return false;
    if (a.lda_bytes >= (1 << 23) || (a.ldw_bytes ? a.ldw_bytes : (long long)a.K * 2) >= (1 << 23)) return false;
    const bool f16io = in_dtype == TP_F16 && out_dtype == TP_F16;
    const bool strided_a = a.rows_per_batch < a.M || a.a_region_s > 0;
    if ((a.flags & TP_LINEAR_NO_STORE) || a.acc_init || a.attn_mode) {
        if (!f16io || strided_a || a.A_parts[0] || ((a.flags & TP_LINEAR_NO_STORE) && a.acc_init)) return false;
        if (a.attn_mode && a.groups != 1) return false;
    }
    if (a.A_parts[0] && out_dtype != TP_F16) return false;
    if (a.a_k_dup && (a.a_k_dup % BK != 0 || 2 * a.a_k_dup > a.K || strided_a || a.A_parts[0] || a.tri)) return false;
    if ((a.flags & TP_LINEAR_ROW_STATS) && out_dtype == TP_F32) return false;
    return in_dtype == TP_BF16 || in_dtype == TP_F16;
}

int gemm_pair_workgroups() { return 2 * gemm8_persistent_cus(); }

static std::atomic<long long> g_pair_launches{0};
long long gemm_pair_launch_count() { return g_pair_launches.load(); }
int gemm_pair_occupancy() {
    auto kern = gemm_pair_kernel<f16_t, f16_t, 0, 0>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES) != hipSuccess) return -1;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), 256, GP_LDS_BYTES) != hipSuccess) return -2;
    return n;
}

template <typename TI, typename TO, int AMODE, int XMODE, int DBG = 0>
static int launch_pair_cfg(const GemmArgs& a, hipStream_t stream) {
#ifdef TP_BUILD_PROBES                                   // (libtokenpacker_exp.so only: the timing probes produce garbage results)
    if constexpr (DBG == 0 && AMODE == 0 && XMODE == 0 && std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value) {
        const int dbg = tuning(TP_TUNE_PAIR_DEBUG) & 7;
        if (dbg == 1) return launch_pair_cfg<TI, TO, AMODE, XMODE, 1>(a, stream);
        if (dbg == 2) return launch_pair_cfg<TI, TO, AMODE, XMODE, 2>(a, stream);
        if (dbg == 3) return launch_pair_cfg<TI, TO, AMODE, XMODE, 3>(a, stream);
        if (dbg == 4) return launch_pair_cfg<TI, TO, AMODE, XMODE, 4>(a, stream);
    }
#endif
    auto kern = gemm_pair_kernel<TI, TO, AMODE, XMODE, DBG>;
    constexpr int lds = GP_LDS_BYTES;
    static DynLdsAttr attr;                             // (per device, a failure is not cached: tp_internal.h)
    const hipError_t attr_err = attr.ensure(reinterpret_cast<const void*>(kern), lds);
    if (attr_err != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", lds, hipGetErrorString(attr_err));
        return TP_ERR_LAUNCH;
    }
    const int tiles_m = (a.M + GP_BM - 1) / GP_BM, tiles_n = a.N / GP_BN;
    const int ntiles = tiles_m * tiles_n;
    int nwg = ntiles;
    int cap = gemm_pair_workgroups();
    if (tuning(TP_TUNE_PAIR_DEBUG) & 8) cap /= 2;       // (probe: one workgroup per CU)
    if (nwg > cap && cap > 0) nwg = cap;
    // stagger of the second workgroup of each CU: half of the period in which the pair turns over two tiles.  A tile is
    // K / 64 K-tiles of 64 MFMAs x 16 cycles per wave; two of them share a SIMD's pipe at ~0.9 utilisation -> a period of
    // ~2300 cycles per K-tile.  One sleep quantum (s_sleep 64) = 4096 cycles.  TP_TUNE_PAIR_STAGGER: percent of that (0 = off).
    int stagger = 0;
    if (ntiles > cap / 2) {
        const long long nk = a.tri ? (a.K / BK) * 5 / 8 : a.K / BK;
        stagger = (int)(nk * 1150 * tuning(TP_TUNE_PAIR_STAGGER) / 100 / 4096);
    }
    dim3 grid((unsigned)nwg, (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, tiles_m, tiles_n, stagger);
    g_pair_launches.fetch_add(1, std::memory_order_relaxed);
    return check_launch("gemm_pair_kernel");
}

template <typename TI, typename TO>
static int launch_pair_types(const GemmArgs& a, hipStream_t stream) {
    const bool strided_a = a.rows_per_batch < a.M || a.a_region_s > 0;
    if ((a.flags & TP_LINEAR_NO_STORE) || a.acc_init || a.attn_mode) {
        if constexpr (std::is_same<TI, f16_t>::value && std::is_same<TO, f16_t>::value) {
            if (a.attn_mode == 1) return launch_pair_cfg<TI, TO, 0, 3>(a, stream);
            if (a.attn_mode == 2) return launch_pair_cfg<TI, TO, 0, 4>(a, stream);
            if (a.flags & TP_LINEAR_NO_STORE) return launch_pair_cfg<TI, TO, 0, 1>(a, stream);
            return launch_pair_cfg<TI, TO, 0, 2>(a, stream);
        }
        set_error("tp gemm pair: NO_STORE / acc_init are built for fp16 operands and output");
        return TP_ERR_INVALID_ARG;
    }
    if (a.A_parts[0]) {
        if constexpr (std::is_same<TO, f16_t>::value) return launch_pair_cfg<TI, TO, 2, 0>(a, stream);
        set_error("tp gemm pair: a multi-part A operand is supported for fp16 output only");
        return TP_ERR_INVALID_ARG;
    }
    return strided_a ? launch_pair_cfg<TI, TO, 1, 0>(a, stream) : launch_pair_cfg<TI, TO, 0, 0>(a, stream);
}

int gemm_pair_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream) {
    if (!gemm_pair_supports(in_dtype, out_dtype, a)) {
        set_error("tp gemm pair: launch not supported by the pair kernel (M=%d N=%d K=%d flags=%d)", a.M, a.N, a.K, a.flags);
        return TP_ERR_INVALID_ARG;
    }
    if (in_dtype == TP_BF16) {
        if (out_dtype == TP_BF16) return launch_pair_types<bf16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch_pair_types<bf16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch_pair_types<bf16_t, float>(a, stream);
    } else {
        if (out_dtype == TP_BF16) return launch_pair_types<f16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch_pair_types<f16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch_pair_types<f16_t, float>(a, stream);
    }
    set_error("tp gemm pair: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
    return TP_ERR_INVALID_ARG;
}

}  // namespace tp
