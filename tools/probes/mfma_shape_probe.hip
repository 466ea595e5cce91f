// mfma_shape_probe.hip — does the MFMA shape matter under MI355X's power limit?  ("revisit 32x32x16")
// Register-resident MFMA loops, no memory traffic: every wave keeps 128 fp32 accumulators (the ping-pong kernel's budget)
// and issues independent MFMAs back to back; 2 waves per SIMD, 256 CUs.  Reports sustained TFLOP/s for
// v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16 on random-ish operands.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_shape_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(512, 2) probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(lane * 4 + i) & 4095]; b[i] = in[(lane * 4 + i + 1777) & 4095]; }
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[32];                                       // 128 accumulator registers
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[8];                                       // 128 accumulator registers
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + 1) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][15];
    }
    if (sum == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = sum;      // keep the loop alive
}

int main() {
    bf16x8* in; float* out;
    hipMalloc(&in, 4096 * sizeof(bf16x8)); hipMalloc(&out, 256 * 512 * sizeof(float));
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // bf16 ~ +-0.01..0.03
    hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;                                  // ~2 ms per launch: long enough for DVFS to settle
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {16, 32}) {
            for (int w = 0; w < 3; ++w) {
                if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(256), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(probe<32>, dim3(256), dim3(512), 0, 0, in, out, iters);
            }
            hipEventRecord(e0);
            const int n = 10;
            for (int w = 0; w < n; ++w) {
                if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(256), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(probe<32>, dim3(256), dim3(512), 0, 0, in, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave and iteration: 16x16x32: 32 MFMAs x 2*16*16*32 flop; 32x32x16: 16 MFMAs x 2*32*32*16 flop — the same 524288
            const double flop = (double)n * 256 * 8 * (double)iters * 524288.0;
            printf("rep %d  v_mfma_f32_%s_bf16: %.3f ms per launch, %.1f TFLOP/s\n", rep, shape == 16 ? "16x16x32" : "32x32x16", ms / n, flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
