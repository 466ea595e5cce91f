// tp_gemm4.hip — 256x256x64 MFMA kernel with ONE wave per SIMD (4 waves, wave tile 128x128) for gfx950:
//   C[M,N] = epilogue( A[M,K] · W[N,K]^T ),  N % 256 == 0.
//
// Same operands, LDS image, fragment layout and fused epilogue as tp_gemm8.hip / tp_gemm.hip (reference
// builder.py:112,113,120,126-130,136).  Why another main loop: on MI355X the large GEMMs of this path run
// POWER-limited (shader clock 1.75-1.85 GHz under load, profiles/r01d), so what counts is work per
// instruction.  A 128x128 wave tile needs 0.25 ds_read_b128 per MFMA instead of 0.375, half the waves, and
// its 256 accumulators live in the AGPR half of the unified 512-register file a lone wave may use.
//
// Software pipeline (per K-tile t, two fragment sets F0/F1 = the two 32-wide k-halves):
//   block A:  64 MFMAs on F0 = k-half 0 of tile t,  interleaved with the 16 ds_read_b128 of F1 = k-half 1 (t)
//   s_waitcnt vmcnt(0) lgkmcnt(0);  s_barrier            (tile t+1 landed + visible; everyone done reading tile t)
//   block B:  64 MFMAs on F1,  interleaved with the 16 LDS-DMA instructions of tile t+2 (into tile t's buffer)
//             and the 16 ds_read_b128 of F0 = k-half 0 of tile t+1
// One barrier per K-tile; the matrix pipe only idles for the barrier skew.  DMA is issued a full K-tile
// (~2000 cycles) before it is waited for.  K and the row-piece advance go through the buffer instruction's
// scalar offset; rows beyond M fall outside the descriptor's range (no access, garbage only feeds rows that
// are never stored).
#include "tp_gemm_common.h"
#include <mutex>

namespace tp {

namespace {
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
#define TP_LAMBDA(arg) [&](auto arg) __attribute__((always_inline))

constexpr int G4_BM = 256, G4_BN = 256, G4_WM = 128, G4_WN = 128;
constexpr int G4_A_BYTES = G4_BM * ROW_BYTES;      // 32 KiB
constexpr int G4_BUF = 2 * G4_A_BYTES;             // 64 KiB: A rows | W rows of one K-tile
constexpr int G4_LDS = 2 * G4_BUF;                 // 128 KiB
}  // namespace

// ILV: 2 = LDS-DMA staging, 3 = register staging; both pin the MFMA / memory interleave by source order
//      (leaving it to hipcc, with or without sched_group_barrier hints, measured 20 % slower: profiles/r01e)
template <typename TI, typename TO, bool STRIDED_A, int ILV>
__global__ void __launch_bounds__(256, 1)
gemm4_kernel(const GemmArgs p, const int tiles_n, const int xcd_swizzle) {
    using X8 = typename Vec<TI>::x8;
    constexpr int BM = G4_BM, BN = G4_BN, WM = G4_WM, WN = G4_WN;
    constexpr int FM = WM / 16, FN = WN / 16;          // 8 x 8 accumulator fragments per wave

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    if (xcd_swizzle) bid = xcd_remap(bid, gridDim.x);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int g = blockIdx.y;

    // ---- buffer descriptors: wave-uniform (kernel arguments, blockIdx, the readfirstlane'd wave id) ----------
    // Wave w stages tile rows 64w..64w+63 of A and of W: eight 1-KiB pieces (8 rows) each per K-tile.
    // rows_per_batch % 64 == 0 (checked on the host), so a wave's 64 A rows never straddle a batch.
    const int a_row0 = m0 + 64 * wave;
    long long a_off;
    if constexpr (STRIDED_A) {
        const int row = a_row0 < p.M ? a_row0 : 0;
        const int b = row / p.rows_per_batch;
        a_off = (long long)b * p.a_batch_stride_bytes + (long long)(row - b * p.rows_per_batch) * p.lda_bytes;
    } else {
        a_off = (long long)a_row0 * p.lda_bytes;
    }
    int a_rows = p.M - a_row0;                          // valid rows of this wave's A span (<= 0: none)
    a_rows = a_rows < 0 ? 0 : (a_rows > 64 ? 64 : a_rows);
    const int a_bytes = a_rows > 0 ? (a_rows - 1) * (int)p.lda_bytes + p.K * 2 : 0;
    const char* a_base = p.A + g * p.a_gs + (a_rows > 0 ? a_off : 0);
    const char* w_base = p.W + g * p.w_gs + (long long)(n0 + 64 * wave) * p.K * 2;
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_bytes, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 64 * p.K * 2, 0x00020000);

    // per-lane source offset inside a piece: row lane/8, 16-B slot (lane%8) ^ (lane/8)  (swizzle on the source)
    const int kslot = (lane & 7) ^ (lane >> 3);
    const int voff_a = (lane >> 3) * (int)p.lda_bytes + kslot * 16;
    const int voff_w = (lane >> 3) * p.K * 2 + kslot * 16;
    const int a_piece = 8 * (int)p.lda_bytes, w_piece = 8 * p.K * 2;

    // one DMA instruction: piece c (0..7 = A, 8..15 = W) of K-tile kt
    auto issue_piece = [&](auto c_, int kt) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value;
        char* dst = smem + (kt & 1) * G4_BUF + (c < 8 ? 0 : G4_A_BYTES) + wave * 8192 + (c & 7) * 1024;
        if constexpr (c < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)dst, 16, voff_a, kt * ROW_BYTES + c * a_piece, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)dst, 16, voff_w, kt * ROW_BYTES + (c - 8) * w_piece, 0, 0);
    };

    // ---- fragment read offsets (swizzled; fragment rows are 16-aligned, so row & 7 == lane & 7) ---------------
    const int slot[2] = {(((lane >> 4)) ^ (lane & 7)) << 4, (((4 + (lane >> 4))) ^ (lane & 7)) << 4};
    const int rd_a = (wm * 128 + (lane & 15)) * ROW_BYTES;                 // + i * 2048
    const int rd_w = G4_A_BYTES + (wn * 128 + (lane & 15)) * ROW_BYTES;    // + j * 2048

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    X8 fa[2][8], fw[2][8];          // [k-half][fragment]

    auto read_frag = [&](auto s_, auto n_, const char* sb) __attribute__((always_inline)) {   // n: 0..7 W, 8..15 A
        constexpr int s = decltype(s_)::value, n = decltype(n_)::value;     // (MFMA order is i-major: W first)
        if constexpr (n < 8) fw[s][n] = *(const X8*)(sb + rd_w + n * 2048 + slot[s]);
        else fa[s][n - 8] = *(const X8*)(sb + rd_a + (n - 8) * 2048 + slot[s]);
    };
    auto mma = [&](auto s_, auto n_) __attribute__((always_inline)) {       // n: 0..63 -> (i, j)
        constexpr int s = decltype(s_)::value, n = decltype(n_)::value;
        constexpr int i = n >> 3, j = n & 7;
        acc[i][j] = Mma<TI>::run(fw[s][j], fa[s][i], acc[i][j]);
    };

    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using T_ = std::true_type; using F_ = std::false_type;
    const int nk = p.K / BK;

    if constexpr (ILV == 2) {
    // ================= LDS-DMA staging, interleave pinned by source order ====================================
    static_for<16>(TP_LAMBDA(c) { issue_piece(c, 0); });
    if (nk > 1) {
        static_for<16>(TP_LAMBDA(c) { issue_piece(c, 1); });
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<16>(TP_LAMBDA(n) { read_frag(S0{}, n, smem); });

    // One K-tile.  MORE: tile t+1 exists;  MORE2: tile t+2 exists.
    auto ktile = [&](auto MORE_, auto MORE2_, const int t) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(MORE_)::value, MORE2 = decltype(MORE2_)::value;
        const char* sb = smem + (t & 1) * G4_BUF;
        const char* sb_next = smem + ((t + 1) & 1) * G4_BUF;
        // ---- block A: k-half 0 of tile t, while k-half 1 streams into F1 --------------------------------------
        static_for<64>(TP_LAMBDA(n) {
            constexpr int nn = decltype(n)::value;
            if constexpr (nn < 32 && nn % 2 == 0) read_frag(S1{}, std::integral_constant<int, nn / 2>{}, sb);
            mma(S0{}, n);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (MORE) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- block B: k-half 1 of tile t, while tile t+2 is requested and k-half 0 of tile t+1 streams into F0 ---
        static_for<64>(TP_LAMBDA(n) {
            constexpr int nn = decltype(n)::value;
            if constexpr (MORE && nn < 32 && nn % 2 == 0) read_frag(S0{}, std::integral_constant<int, nn / 2>{}, sb_next);
            if constexpr (MORE2 && nn >= 32 && nn % 2 == 0) issue_piece(std::integral_constant<int, (nn - 32) / 2>{}, t + 2);
            mma(S1{}, n);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int t = 0;
    for (; t < nk - 2; ++t) ktile(T_{}, T_{}, t);
    if (nk >= 2) { ktile(T_{}, F_{}, t); ++t; }
    ktile(F_{}, F_{}, t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    } else {
    // ================= register staging (global -> VGPR -> ds_write_b128), hipBLASLt-style =====================
    // A lone wave cannot hide the ~60-cycle issue cost of an LDS-DMA instruction behind a partner's MFMAs; plain
    // buffer loads and ds_write_b128 issue in a few cycles each.  One staging set of 16 x 16 B per lane:
    //   block A(t):  ds_write stage[c] -> LDS buffer of tile t+1 (c = 0..15), each followed (c < 8) by the load of
    //                piece c of tile t+2 into the same registers;  16 ds_read of F1 = k-half 1 (t)
    //   barrier      (tile t+1 written + visible, everyone done reading tile t-1's buffer)
    //   block B(t):  16 ds_read of F0 = k-half 0 (t+1);  loads of pieces 8..15 of tile t+2
    // so a global load has a whole K-tile (~2000 cycles) before its ds_write needs it.
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t stage[16];
    auto gload = [&](auto c_, int kt) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value;
        if constexpr (c < 8)
            stage[c] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a, kt * ROW_BYTES + c * a_piece, 0));
        else
            stage[c] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff_w, kt * ROW_BYTES + (c - 8) * w_piece, 0));
    };
    auto lwrite = [&](auto c_, int kt) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value;
        char* dst = smem + (kt & 1) * G4_BUF + (c < 8 ? 0 : G4_A_BYTES) + wave * 8192 + (c & 7) * 1024 + lane * 16;
        *(u32x4_t*)dst = stage[c];
    };
    static_for<16>(TP_LAMBDA(c) { gload(c, 0); });
    static_for<16>(TP_LAMBDA(c) { lwrite(c, 0); });
    if (nk > 1) static_for<16>(TP_LAMBDA(c) { gload(c, 1); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<16>(TP_LAMBDA(n) { read_frag(S0{}, n, smem); });

    auto ktile = [&](auto MORE_, auto MORE2_, const int t) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(MORE_)::value, MORE2 = decltype(MORE2_)::value;
        const char* sb = smem + (t & 1) * G4_BUF;
        const char* sb_next = smem + ((t + 1) & 1) * G4_BUF;
        static_for<64>(TP_LAMBDA(n) {
            constexpr int nn = decltype(n)::value;
            if constexpr (MORE && nn % 4 == 0) lwrite(std::integral_constant<int, nn / 4>{}, t + 1);
            if constexpr (MORE2 && nn % 4 == 1 && nn / 4 < 8) gload(std::integral_constant<int, nn / 4>{}, t + 2);
            if constexpr (nn % 4 == 2) read_frag(S1{}, std::integral_constant<int, nn / 4>{}, sb);
            mma(S0{}, n);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (MORE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<64>(TP_LAMBDA(n) {
            constexpr int nn = decltype(n)::value;
            if constexpr (MORE && nn < 32 && nn % 2 == 0) read_frag(S0{}, std::integral_constant<int, nn / 2>{}, sb_next);
            if constexpr (MORE2 && nn >= 32 && nn % 4 == 0) gload(std::integral_constant<int, 8 + (nn - 32) / 4>{}, t + 2);
            mma(S1{}, n);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int t = 0;
    for (; t < nk - 2; ++t) ktile(T_{}, T_{}, t);
    if (nk >= 2) { ktile(T_{}, F_{}, t); ++t; }
    ktile(F_{}, F_{}, t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    float2 mean_rstd[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        mean_rstd[i] = make_float2(0.f, 1.f);
        if (p.flags & TP_LINEAR_LN_FOLD) {
            int m = m0 + wm * WM + i * 16 + (lane & 15);
            m = m < p.M ? m : p.M - 1;
            mean_rstd[i] = *(const float2*)(p.stats_in + g * p.stats_in_gs + (long long)m * 2);
        }
    }
    gemm_epilogue<TO, BM, BN, WM, WN>(acc, p, g, m0, n0, tile_n, wm, wn, lane, tid, mean_rstd, smem);
}

// ---- host side ------------------------------------------------------------------------------------------
template <typename TI, typename TO, bool STRIDED_A, int ILV>
static int launch4_cfg(const GemmArgs& a, hipStream_t stream) {
    auto kern = gemm4_kernel<TI, TO, STRIDED_A, ILV>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS);
    });
    if (attr_err != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", G4_LDS, hipGetErrorString(attr_err));
        return TP_ERR_LAUNCH;
    }
    const int tiles_m = (a.M + G4_BM - 1) / G4_BM, tiles_n = a.N / G4_BN;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)a.groups, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), G4_LDS, stream, a, tiles_n, tuning(TP_TUNE_XCD_SWIZZLE));
    return check_launch("gemm4_kernel");
}

template <typename TI, typename TO, int ILV>
static int launch4_var(const GemmArgs& a, hipStream_t stream) {
    return a.rows_per_batch < a.M ? launch4_cfg<TI, TO, true, ILV>(a, stream)
                                  : launch4_cfg<TI, TO, false, ILV>(a, stream);
}

template <typename TI, typename TO>
static int launch4_types(const GemmArgs& a, hipStream_t stream) {
    switch (tuning(TP_TUNE_GEMM_KERNEL)) {
        case 13: return launch4_var<TI, TO, 3>(a, stream);
        default: return launch4_var<TI, TO, 2>(a, stream);
    }
}

bool gemm4_supports(const GemmArgs& a) {
    if (a.N % G4_BN != 0 || a.K % BK != 0) return false;
    if (a.rows_per_batch < a.M && a.rows_per_batch % 64 != 0) return false;
    if ((long long)63 * a.lda_bytes + (long long)a.K * 2 > 0x7fffffffLL) return false;
    return true;
}

int gemm4_launch(int in_dtype, int out_dtype, const GemmArgs& a, hipStream_t stream) {
    if (in_dtype == TP_BF16) {
        if (out_dtype == TP_BF16) return launch4_types<bf16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch4_types<bf16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch4_types<bf16_t, float>(a, stream);
    } else if (in_dtype == TP_F16) {
        if (out_dtype == TP_BF16) return launch4_types<f16_t, bf16_t>(a, stream);
        if (out_dtype == TP_F16) return launch4_types<f16_t, f16_t>(a, stream);
        if (out_dtype == TP_F32) return launch4_types<f16_t, float>(a, stream);
    }
    set_error("tp gemm4: unsupported dtypes in=%d out=%d", in_dtype, out_dtype);
    return TP_ERR_INVALID_ARG;
}

}  // namespace tp
