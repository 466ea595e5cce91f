// operand_fetch_probe.hip — what does ONE CU's L2 -> LDS operand path deliver, by instruction?  (the gemm8 main loop
// fetches its operands by LDS-DMA at ~62 GB/s per CU with 48 KiB in flight, profiles/r04m_loop_probe.json; the vendor library stages its
// prefetch in registers.  Before building a register-staged fetch into the 8-wave layout: is that path faster at all?)
//
// 256 workgroups x 8 waves (one per CU, 128 KiB of LDS each, like the ping-pong kernel), every workgroup streams ITS 64-KiB window of
// an L2-resident buffer (2 MiB per XCD) again and again — 64 KiB = the operand bytes of one 256 x 256 x 64 K-tile, as 64 pieces of 1 KiB
// (8 per wave), 16 B per lane, whole 128-B lines.  Modes:
//   0  buffer_load_dwordx4 ... lds           (LDS-DMA, the kernels' fetch), counted s_waitcnt vmcnt(IN_FLIGHT - 1) behind every piece
//   1  global_load_dwordx4 -> VGPR           (register path, the data is only consumed by an empty asm: the L1 / TA side alone)
//   2  global_load_dwordx4 -> VGPR -> ds_write_b128   (register-staged fetch: what hipBLASLt-style kernels do)
// IN_FLIGHT = pieces per wave the pipeline keeps outstanding (4, 8 or 12: 32 / 64 / 96 KiB per CU).
// PATTERN (mode 0): 0 a piece = 1 KiB contiguous | 1 a piece = 8 rows x 128 B, rows 8 KiB apart (a K-tile slab of row-major operands with
// K = 4096: what the GEMM's DMA instructions actually address) | 2 = 1 + the kernels' XOR swizzle of the 16-B slots inside a row.  The
// LINES touched are the same 64 KiB per workgroup in every pattern (L2-resident); only the address pattern of an instruction changes.
// Prints GB/s per CU and in total; modes are interleaved over three repetitions.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/operand_fetch_probe.hip -o /tmp/fetch_probe && /tmp/fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
}

template <int MODE, int IN_FLIGHT, bool BARRIER, int PATTERN = 0>
__global__ void __launch_bounds__(512, 2) probe(const char* __restrict__ src, float* __restrict__ out, int iters, int stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD x = bid % 8 owns windows [x * 32, x * 32 + 32): 2 MiB per XCD, L2-resident after the first sweep
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const char* win = src + (PATTERN == 0 ? ((size_t)(x * 32 + slot) << 16) : ((size_t)(x * 32 + slot) * (4 << 20) + (x * 32 + slot) * 65536ull % (1 << 20)));   // (rows 8 KiB apart: > 4 MiB of span)
    // piece p (0..7) of this wave: bytes [(p * 8 + wave) * 1024, + 1024) of the window, lane l its 16 B — rows of 128 B, like a K-tile
    int voff = wave * 1024 + lane * 16;
    // PATTERN >= 1: a piece = 8 rows x 128 B of a row-major operand.  LDS position of lane l: row l / 8 of the piece, 16-B slot l % 8.
    //   1 identity          2 slot ^= row (the kernels' swizzle)      3 slot = (slot + row) % 8 (rotation)
    //   4 identity, the piece's 8 rows are 8 rows apart (row = 8 i + j instead of 8 j + i)      5 / 6: 1 / 2 with a row stride of 8 KiB + 128 B
    //   7 slot ^= row & 4 (half-line swap only)      8 slot ^= row & 3 (permutation inside a 64-B half only)
    constexpr int LDA = (PATTERN == 5 || PATTERN == 6) ? 8192 + 128 : 8192;
    if constexpr (PATTERN >= 1 && PATTERN < 9) {
        const int r = lane >> 3, pslot = lane & 7;
        const int slot16 = (PATTERN == 2 || PATTERN == 6) ? (pslot ^ r) : PATTERN == 3 ? ((pslot + r) & 7) : PATTERN == 7 ? (pslot ^ (r & 4)) :
                           PATTERN == 8 ? (pslot ^ (r & 3)) : pslot;
        const int row = PATTERN == 4 ? (r * 8 + wave) : (wave * 8 + r);          // of the 64-row block a piece index p selects
        voff = row * LDA + slot16 * 16;
    }
    constexpr int PSTEP = PATTERN == 0 ? 8192 : 64 * LDA;           // piece p -> p-th block of 64 rows
    if constexpr (MODE == 0 && PATTERN >= 13) {
        // (round 6) Shared STRIDED streams — the GEMM's real operand walk: a step = one 64-wide K-tile slab of 512 operand rows (64 pieces of
        // 8 rows x 128 B, rows LDS apart), 64 steps along K (128 B each, K = 4096), then the next 512 rows; the workgroups of an XCD all read
        // the same stream.  13: row stride 8 KiB (K = 4096 fp16, what the GEMMs address)  14: 8 KiB + 128 B  15: 2 KiB (K = 1024)  16: 2 KiB + 128 B
        constexpr int LDS_ = (PATTERN == 13 ? 8192 : PATTERN == 14 ? 8192 + 128 : PATTERN == 15 ? 2048 : 2048 + 128);
        constexpr int KSTEPS = (PATTERN <= 14 ? 64 : 16);
        const char* base = src + (size_t)x * (100u << 20);               // 100 MiB of span per XCD: 24 blocks of 512 rows
        const int r = lane >> 3, pslot = lane & 7;
        const int voff_s = (wave * 8 + r) * LDS_ + pslot * 16;          // piece p of this wave: rows 64 p + 8 wave + r
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        int kt = 0, blk = 0;
        for (int it = 0; it < iters; ++it) {
            char* dst = smem + (it & 1) * 65536 + wave * 1024;
            const int soff = blk * 512 * LDS_ + kt * 128;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + p * 8192), 16, voff_s, soff + p * 64 * LDS_, 0, 0);
                wait_vm<IN_FLIGHT - 1>();
            }
            if (++kt == KSTEPS) { kt = 0; blk = blk + 1 == 24 ? 0 : blk + 1; }
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 0 && PATTERN >= 9) {
        // Shared streams (what a GEMM's operands are): the 32 workgroups of an XCD read the SAME 64 KiB per step, in lockstep only by
        // their equal pace.  9: always the same 64 KiB (L2-resident, shared)  10: a 24-MiB region per XCD walked round and round
        // (beyond the 4-MiB L2, inside the Infinity Cache)  11: as 10 but each workgroup starts a quarter of the region apart in
        // groups of 8 (four streams per XCD, 8 sharers each — kv_layer0's A panels)  12: a PRIVATE 24-MiB-per-XCD / 32 stream each
        const char* base = src + (size_t)x * (24u << 20);
        const int steps = (24 << 20) / 65536;                            // 384 steps per lap
        int st = PATTERN == 11 ? (slot >> 3) * (steps / 4) : PATTERN == 12 ? slot * (steps / 32) : 0;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        for (int it = 0; it < iters; ++it) {
            char* dst = smem + (it & 1) * 65536 + wave * 1024;
            const int soff = PATTERN == 9 ? 0 : st * 65536;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + p * 8192), 16, voff, soff + p * 8192, 0, 0);
                wait_vm<IN_FLIGHT - 1>();
            }
            st = st + 1 == steps ? 0 : st + 1;
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 0) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, 0x7fffffff, 0x00020000);
        for (int it = 0; it < iters; ++it) {
            char* dst = smem + (it & 1) * 65536 + wave * 1024;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + p * 8192), 16, voff, p * PSTEP, 0, 0);
                wait_vm<IN_FLIGHT - 1>();
            }
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // software pipeline over pieces: ring[k] holds the load issued IN_FLIGHT pieces ago
        u32x4 ring[IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < IN_FLIGHT; ++k) ring[k] = *(const u32x4*)(win + voff + (k & 7) * 8192);
        // (`stride` is 0 at run time: the addresses repeat, but the compiler cannot know and hoist the loads out of the loop)
        static_assert(IN_FLIGHT == 4 || IN_FLIGHT == 8 || IN_FLIGHT == 12, "ring sizes that keep the unrolled indices static");
        constexpr int UNROLL = IN_FLIGHT == 12 ? 24 : 8;            // a multiple of both 8 (pieces) and IN_FLIGHT
        for (int it = 0; it < iters; it += UNROLL / 8) {
#pragma unroll
            for (int q = 0; q < UNROLL; ++q) {
                const int p = q & 7;
                const u32x4 v = ring[q % IN_FLIGHT];
                ring[q % IN_FLIGHT] = *(const u32x4*)(win + voff + p * 8192 + (((it + q / 8) * stride) & 0x1ff0));
                if constexpr (MODE == 2) *(u32x4*)(smem + ((it + q / 8) & 1) * 65536 + wave * 1024 + p * 8192 + lane * 16) = v;
                else asm volatile("" :: "v"(v));
                if constexpr (BARRIER) if (p == 7) __builtin_amdgcn_s_barrier();
            }
        }
#pragma unroll
        for (int k = 0; k < IN_FLIGHT; ++k) asm volatile("" :: "v"(ring[k]));
    }
    __syncthreads();
    if (iters < 0) out[blockIdx.x * 512 + threadIdx.x] = *(const float*)(smem + threadIdx.x * 4);     // keep the LDS writes alive
}

static int g_nwg = 256;          // argv[1]: workgroups (256 = one per CU; 64 = 8 per XCD, round 6: is the fetch rate a per-CU or a per-XCD figure?)
template <int MODE, int IN_FLIGHT, bool BARRIER, int PATTERN = 0>
static double run(const char* src, float* out, int iters) {
    auto kern = probe<MODE, IN_FLIGHT, BARRIER, PATTERN>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(g_nwg), dim3(512), 131072, 0, src, out, iters / 8, 0);    // warm-up: the windows reach the L2s
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(g_nwg), dim3(512), 131072, 0, src, out, iters, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)iters * 65536.0 / (ms * 1e-3) / 1e9;       // GB/s per CU
}

int main(int argc, char** argv) {
    if (argc > 1) g_nwg = atoi(argv[1]);
    const bool quick = argc > 2;                               // argv[2]: only the LDS-DMA rows
    printf("workgroups: %d (%d per XCD)\n", g_nwg, g_nwg / 8);
    char* src; float* out;
    const size_t bytes = ((size_t)256 << 22) + (64u << 20);                   // 1 GiB of span for the row patterns (64 KiB of lines per workgroup are touched)
    hipMalloc(&src, bytes); hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMemset(src, 1, bytes);
    const int iters = 24000;                                   // 1.5 GB per CU: ~15-30 ms per launch
    for (int rep = 0; rep < (quick ? 2 : 3); ++rep) {
        printf("rep %d  (GB/s per CU; x256 = chip)\n", rep);
#define ROW(MODE, NAME) \
        printf("  %-44s  in flight 4: %6.1f   8: %6.1f   12: %6.1f   | with a barrier per 64 KiB, 8 in flight: %6.1f\n", NAME, \
               run<MODE, 4, false>(src, out, iters), run<MODE, 8, false>(src, out, iters), run<MODE, 12, false>(src, out, iters), \
               run<MODE, 8, true>(src, out, iters));
        ROW(0, "buffer_load_dwordx4 ... lds (LDS-DMA)")
        if (!quick) {
        ROW(1, "global_load_dwordx4 -> VGPR (no LDS write)")
        ROW(2, "global_load_dwordx4 -> VGPR -> ds_write_b128")
        }
#define PAT(P, NAME) printf("    %-58s  in flight 6: %6.1f   8: %6.1f   12: %6.1f   | barrier per 64 KiB, 6 in flight: %6.1f\n", NAME, \
               run<0, 6, false, P>(src, out, iters), run<0, 8, false, P>(src, out, iters), run<0, 12, false, P>(src, out, iters), run<0, 6, true, P>(src, out, iters));
        printf("  LDS-DMA by address pattern of a piece:\n");
        PAT(0, "1 KiB contiguous")
        PAT(1, "8 rows x 128 B (8 KiB apart), slots in order")
        PAT(2, "8 rows x 128 B, slot ^= row  (the kernels' swizzle)")
        PAT(3, "8 rows x 128 B, slot = (slot + row) % 8")
        PAT(7, "8 rows x 128 B, slot ^= row & 4 (64-B halves swapped)")
        PAT(8, "8 rows x 128 B, slot ^= row & 3 (inside a 64-B half)")
        PAT(4, "8 rows x 128 B, slots in order, rows 64 KiB apart")
        PAT(5, "8 rows x 128 B, slots in order, row stride 8 KiB + 128 B")
        PAT(6, "8 rows x 128 B, slot ^= row, row stride 8 KiB + 128 B")
        printf("  LDS-DMA, 1-KiB pieces, streams SHARED inside an XCD (the 32 workgroups read the same bytes at their own pace):\n");
        PAT(9, "the same 64 KiB for ever (L2-resident, 32 sharers)")
        PAT(10, "one 24-MiB stream per XCD (Infinity Cache), 32 sharers")
        PAT(11, "four 24-MiB streams per XCD, 8 sharers each")
        PAT(12, "32 streams per XCD, no sharing")
        printf("  LDS-DMA, the GEMM's own walk (8 rows x 128 B per piece, K-tile after K-tile along the rows), one stream per XCD, all its workgroups share it:\n");
        PAT(13, "row stride 8 KiB (K = 4096)")
        PAT(14, "row stride 8 KiB + 128 B")
        PAT(15, "row stride 2 KiB (K = 1024)")
        PAT(16, "row stride 2 KiB + 128 B")
        fflush(stdout);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
