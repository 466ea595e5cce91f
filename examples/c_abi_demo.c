/* c_abi_demo.c — the drop-in boundary used from plain C (no Python, no torch): what a maintainer binding
 * libtokenpacker_hip.so from another host language would write.
 *
 *   gcc -O2 -std=c11 examples/c_abi_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Ltokenpacker_amd \
 *       -ltokenpacker_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/tokenpacker_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/tp_demo
 *   /tmp/tp_demo <dir> <B> <scale_factor> <hidden_size> <dtype: 0 bf16 | 1 fp16>
 *
 * <dir> holds the 23 reference state-dict tensors as raw 16-bit files named after tp_weights' fields
 * (q_proj_1_weight.bin ...), x.bin [B,576,1024] and x_multi.bin [B,576,4096]; the result [B,M,D] is written to
 * <dir>/out.bin.  tests/test_gpu_capi_c.py checks it bit for bit against tokenpacker_amd.TokenPacker on the same files.
 * Replaces, for that caller, `TokenPacker.__init__ + load_state_dict + forward` (builder.py:40-137). */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tokenpacker.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_TP(e) do { int r_ = (e); if (r_ != TP_OK) { fprintf(stderr, "tp: %d %s (%s:%d)\n", r_, tp_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static void* upload(const char* dir, const char* name, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(4); }
    void* host = malloc(bytes);
    if (fread(host, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read (want %zu bytes)\n", path, bytes); exit(4); }
    fclose(f);
    void* dev = NULL;
    if (hipMalloc(&dev, bytes) != hipSuccess || hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload of %s failed\n", name); exit(4); }
    free(host);
    return dev;
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s <dir> <B> <scale_factor> <hidden_size> <dtype>\n", argv[0]); return 1; }
    const char* dir = argv[1];
    const int B = atoi(argv[2]), s = atoi(argv[3]), D = atoi(argv[4]), dt = atoi(argv[5]);
    const size_t E = 1024, C4 = 4096, N = 576, es = 2;
    if (tp_version() != TP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    tp_desc d = {B, 24, s, D, dt, dt, 1e-6f, 0};
    tp_desc bad = d; bad.scale_factor = 5;                       /* the reference's ValueError (builder.py:51-52) as a status */
    if (tp_workspace_bytes(&bad) != 0 || strstr(tp_last_error(), "scale_factor must be divisible by grid size") == NULL) { fprintf(stderr, "bad scale not refused\n"); return 1; }

    tp_weights w;
    w.q_proj_1_weight = upload(dir, "q_proj_1_weight", E * E * es);
    w.k_proj_1_0_weight = upload(dir, "k_proj_1_0_weight", E * C4 * es);  w.k_proj_1_0_bias = upload(dir, "k_proj_1_0_bias", E * es);
    w.k_proj_1_2_weight = upload(dir, "k_proj_1_2_weight", E * E * es);   w.k_proj_1_2_bias = upload(dir, "k_proj_1_2_bias", E * es);
    w.v_proj_1_0_weight = upload(dir, "v_proj_1_0_weight", E * C4 * es);  w.v_proj_1_0_bias = upload(dir, "v_proj_1_0_bias", E * es);
    w.v_proj_1_2_weight = upload(dir, "v_proj_1_2_weight", E * E * es);   w.v_proj_1_2_bias = upload(dir, "v_proj_1_2_bias", E * es);
    w.ln_q_1_weight = upload(dir, "ln_q_1_weight", E * es);  w.ln_q_1_bias = upload(dir, "ln_q_1_bias", E * es);
    w.ln_k_1_weight = upload(dir, "ln_k_1_weight", E * es);  w.ln_k_1_bias = upload(dir, "ln_k_1_bias", E * es);
    w.ln_v_1_weight = upload(dir, "ln_v_1_weight", E * es);  w.ln_v_1_bias = upload(dir, "ln_v_1_bias", E * es);
    w.clip_attn_in_proj_weight = upload(dir, "clip_attn_in_proj_weight", 3 * E * E * es);
    w.clip_attn_in_proj_bias = upload(dir, "clip_attn_in_proj_bias", 3 * E * es);
    w.clip_attn_out_proj_weight = upload(dir, "clip_attn_out_proj_weight", E * E * es);
    w.clip_attn_out_proj_bias = upload(dir, "clip_attn_out_proj_bias", E * es);
    w.mlp_0_weight = upload(dir, "mlp_0_weight", (size_t)D * E * es);  w.mlp_0_bias = upload(dir, "mlp_0_bias", (size_t)D * es);
    w.mlp_2_weight = upload(dir, "mlp_2_weight", (size_t)D * D * es);  w.mlp_2_bias = upload(dir, "mlp_2_bias", (size_t)D * es);
    void* x = upload(dir, "x", (size_t)B * N * E * es);
    void* xm = upload(dir, "x_multi", (size_t)B * N * C4 * es);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    const size_t pbytes = tp_packed_weight_bytes(&d), wbytes = tp_workspace_bytes(&d);
    const size_t M = (size_t)(24 / s) * (24 / s), obytes = (size_t)B * M * D * es;
    void *packed, *ws, *out, *out2;
    CHECK_HIP(hipMalloc(&packed, pbytes));  CHECK_HIP(hipMalloc(&ws, wbytes));
    CHECK_HIP(hipMalloc(&out, obytes));     CHECK_HIP(hipMalloc(&out2, obytes));
    CHECK_HIP(hipMemsetAsync(ws, 0, TP_WORKSPACE_STATUS_BYTES, stream));   /* the status block: zeroed ONCE by the caller */
    CHECK_TP(tp_pack_weights(&d, &w, packed, pbytes, stream));
    int32_t status[3] = {-1, -1, -1};
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(status, (char*)packed + tp_packed_status_offset(&d), sizeof status, hipMemcpyDeviceToHost));
    if (status[0] != 0) { fprintf(stderr, "%d weight elements exceed the fp16 range\n", status[0]); return 5; }

    const int64_t xs[3] = {(int64_t)(N * E), (int64_t)E, 1}, xms[3] = {(int64_t)(N * C4), (int64_t)C4, 1};
    CHECK_TP(tp_forward(&d, x, xs, xm, xms, packed, out, ws, wbytes, stream));
    CHECK_TP(tp_forward(&d, x, xs, xm, xms, packed, out2, ws, wbytes, stream));      /* enqueue-only: no sync in between */
    int32_t* sat;
    CHECK_HIP(hipMalloc((void**)&sat, TP_NUM_DEBUG_BUFFERS * sizeof(int32_t)));
    CHECK_TP(tp_debug_count_saturated(&d, ws, wbytes, sat, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    uint16_t* h1 = (uint16_t*)malloc(obytes); uint16_t* h2 = (uint16_t*)malloc(obytes);
    int32_t hs[TP_NUM_DEBUG_BUFFERS], sticky = -1;
    CHECK_HIP(hipMemcpy(&sticky, ws, sizeof sticky, hipMemcpyDeviceToHost));   /* sticky fp16-saturation bits of the two forwards */
    if (sticky != 0) { fprintf(stderr, "an epilogue clamped (stage bits 0x%x)\n", sticky); return 8; }
    CHECK_HIP(hipMemcpy(h1, out, obytes, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h2, out2, obytes, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hs, sat, sizeof hs, hipMemcpyDeviceToHost));
    if (memcmp(h1, h2, obytes) != 0) { fprintf(stderr, "two forwards differ: not deterministic\n"); return 6; }
    long saturated = 0;
    for (int i = 0; i < TP_NUM_DEBUG_BUFFERS; ++i) saturated += hs[i];
    char path[1024];
    snprintf(path, sizeof path, "%s/out.bin", dir);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(h1, 1, obytes, f) != obytes) { fprintf(stderr, "cannot write %s\n", path); return 7; }
    fclose(f);
    printf("c_abi_demo: B=%d s=%d D=%d dtype=%d -> out [%d, %zu, %d], deterministic, saturated=%ld, fold=%d fused_ln=%d\n",
           B, s, D, dt, B, M, D, saturated, status[1], status[2]);
    return 0;
}
