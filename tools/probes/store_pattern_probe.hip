// store_pattern_probe.hip — what does ONE CU's store path deliver, by the address pattern of a buffer_store wave-instruction?  (round 6)
//
// The persistent GEMM's epilogue spends 5.3 us per 256 x 256 tile in its 128 output stores (16 per wave, ~65-75 cycles each), and
// tools/probes/store_bound_probe.py shows that cost per CU is the same with 32 or 256 CUs active: a per-CU bound, not HBM's.  The
// epilogue's stores put lane l at (row l & 15, 16-B chunk l >> 4) of a 16-row x 64-B block: ADJACENT LANES ARE ADJACENT ROWS (8 KiB
// apart), the four lanes that share a 64-B segment are 16 lanes apart.  hipBLASLt's kernel stores 4 rows x 256 B with adjacent lanes
// adjacent in memory.  This probe issues the same bytes per instruction (1 KiB, dwordx4) under different lane -> address maps:
//   0  row = l & 15, chunk = l >> 4        16 rows x 64 B, lane-adjacent rows          (the GEMM epilogue today)
//   1  row = l >> 2, chunk = l & 3         16 rows x 64 B, lane-adjacent chunks        (same lines, lanes permuted)
//   2  row = l >> 3, chunk = l & 7          8 rows x 128 B (whole lines), lane-adjacent chunks
//   3  row = l >> 4, chunk = l & 15         4 rows x 256 B                              (the vendor kernel's stores)
//   4  row = l & 7,  chunk = l >> 3         8 rows x 128 B, lane-adjacent ROWS
//   5  1 KiB contiguous
// x {plain, nt} x {8 waves, 4 waves per workgroup}; 256 workgroups (one per CU; 140 KiB of LDS requested so that no second one fits) or
// 32.  Every workgroup writes "tiles" of 256 x 256 f16 (128 KiB) into its own rows of a [M, 4096] f16 matrix (row stride 8 KiB), as the
// GEMM does: wave w of 8 owns rows 128 (w / 4) .. + 127 and columns 64 (w % 4) .. + 63 of the tile.  Prints us per tile, cycles per
// store instruction (at the measured time and a nominal 2.0 GHz) and GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool NT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) probe(char* __restrict__ out, int tiles, long long ldc_bytes, int tiles_n) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int row, chunk;
    if (PATTERN == 0) { row = lane & 15; chunk = lane >> 4; }
    else if (PATTERN == 1) { row = lane >> 2; chunk = lane & 3; }
    else if (PATTERN == 2) { row = lane >> 3; chunk = lane & 7; }
    else if (PATTERN == 3) { row = lane >> 4; chunk = lane & 15; }
    else if (PATTERN == 4) { row = lane & 7; chunk = lane >> 3; }
    else { row = 0; chunk = lane; }
    // rows / bytes one instruction covers
    constexpr int IROWS = PATTERN == 0 || PATTERN == 1 ? 16 : PATTERN == 2 || PATTERN == 4 ? 8 : PATTERN == 3 ? 4 : 1;
    constexpr int IBYTES = 1024 / IROWS;
    // the wave's block of the tile: 8 waves: 128 rows x 128 B; 4 waves: 128 rows x 256 B (columns 128 (w % 2))
    constexpr int WBYTES = WAVES == 8 ? 128 : 256;
    constexpr int WROWS = 128;
    const int wr = WAVES == 8 ? wave >> 2 : wave >> 1, wc = WAVES == 8 ? wave & 3 : wave & 1;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 0x3c003c00u, 0x3c003c00u};
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const long long base = (long long)(tm * 256 + wr * WROWS) * ldc_bytes + tn * 512 + wc * WBYTES;
        const unsigned sbase = (unsigned)base;             // (< 2 GiB by construction)
        // instructions of this wave's block: column groups of IBYTES (if IBYTES < WBYTES) x row groups of IROWS
#pragma unroll 4
        for (int r0 = 0; r0 < WROWS; r0 += IROWS) {
#pragma unroll
            for (int c0 = 0; c0 < WBYTES; c0 += (IBYTES < WBYTES ? IBYTES : WBYTES)) {
                if constexpr (IBYTES > WBYTES) {
                    // (patterns 3, 5 with 8 waves: an instruction is wider than the wave's block: fold the excess into more rows)
                    constexpr int F = IBYTES / WBYTES;
                    const int rr = row * F + (chunk * 16) / WBYTES, cc = (chunk * 16) % WBYTES;
                    const unsigned voff = (unsigned)((r0 * F + rr) * ldc_bytes) + cc;
                    if (r0 * F < WROWS) {
                        if constexpr (NT) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" :: "v"(v), "v"(voff), "s"(rsrc), "s"(sbase) : "memory");
                        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(voff), "s"(rsrc), "s"(sbase) : "memory");
                    }
                } else {
                    const unsigned voff = (unsigned)((r0 + row) * ldc_bytes) + c0 + chunk * 16;
                    if constexpr (NT) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" :: "v"(v), "v"(voff), "s"(rsrc), "s"(sbase) : "memory");
                    else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(voff), "s"(rsrc), "s"(sbase) : "memory");
                }
            }
        }
        v.z += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PATTERN, bool NT, int WAVES>
static float run(char* out, int nwg, int tiles, hipStream_t s) {
    auto k = probe<PATTERN, NT, WAVES>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); once = true; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(nwg), dim3(WAVES * 64), 140 * 1024, s, out, tiles, 8192LL, 16);
    CK(hipStreamSynchronize(s));
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k, dim3(nwg), dim3(WAVES * 64), 140 * 1024, s, out, tiles, 8192LL, 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}

template <int PATTERN, bool NT, int WAVES>
static void report(char* out, hipStream_t s, const char* name) {
    for (int nwg : {256, 32}) {
        const int tiles = 64;                          // per workgroup: 64 x 128 KiB = 8 MiB; 256 workgroups: 2 GiB
        // launch overhead cancels in the difference of two tile counts
        const float t1 = run<PATTERN, NT, WAVES>(out, nwg, tiles / 4, s), t2 = run<PATTERN, NT, WAVES>(out, nwg, tiles, s);
        const double us_tile = (t2 - t1) * 1e3 / (tiles - tiles / 4);
        const double instr_per_cu_tile = 128.0;
        printf("pattern %d %-44s %s waves %d  wgs %3d : %6.3f us / tile  = %5.1f ns per store instruction and CU (%4.0f cycles at 2.0 GHz)  %6.1f GB/s per CU  %6.2f TB/s total\n",
               PATTERN, name, NT ? "nt   " : "plain", WAVES, nwg, us_tile, us_tile * 1e3 / instr_per_cu_tile, us_tile * 1e3 / instr_per_cu_tile * 2.0,
               131072.0 / us_tile * 1e-3, 131072.0 * nwg / us_tile * 1e-6);
    }
    fflush(stdout);
}

int main() {
    char* out; CK(hipMalloc(&out, (size_t)2200 << 20));
    CK(hipMemset(out, 0, (size_t)2200 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        printf("== repetition %d\n", rep);
        report<0, false, 8>(out, s, "16 rows x 64 B, lane-adjacent ROWS (GEMM today)");
        report<1, false, 8>(out, s, "16 rows x 64 B, lane-adjacent chunks");
        report<2, false, 8>(out, s, "8 rows x 128 B, lane-adjacent chunks");
        report<4, false, 8>(out, s, "8 rows x 128 B, lane-adjacent ROWS");
        report<3, false, 8>(out, s, "4 rows x 256 B (folded: 8 rows x 128 B)");
        report<5, false, 8>(out, s, "1 KiB contiguous (folded: 8 rows x 128 B)");
        report<0, true, 8>(out, s, "16 rows x 64 B, lane-adjacent ROWS (GEMM today)");
        report<1, true, 8>(out, s, "16 rows x 64 B, lane-adjacent chunks");
        report<2, true, 8>(out, s, "8 rows x 128 B, lane-adjacent chunks");
        report<0, false, 4>(out, s, "16 rows x 64 B, lane-adjacent ROWS");
        report<2, false, 4>(out, s, "8 rows x 128 B, lane-adjacent chunks");
        report<3, false, 4>(out, s, "4 rows x 256 B (vendor)");
        report<3, true, 4>(out, s, "4 rows x 256 B (vendor)");
    }
    return 0;
}
