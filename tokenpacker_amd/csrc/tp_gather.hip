// tp_gather.hip — the all-gather of projected tokens without a collective kernel (the reference has no
// counterpart: it never calls torch.distributed).
//
// One process per GPU.  Every rank owns `depth` receive buffers [total, M, D]; its projector writes the local shard
// straight into rows [lo, hi) of the buffer of the step (TokenPacker.forward(..., _out=...)), and the shard then travels
// to the SAME rows of every peer's buffer as one hipMemcpyAsync per peer over that peer's xGMI link — SDMA engines
// (hipMemcpyDeviceToDeviceNoCU), zero compute units: RCCL's all-gather kernels want CUs, and the persistent GEMMs of the
// NEXT forward own every CU (one 512-thread workgroup, 133-156 KiB of LDS and the whole register file each).
//
// Synchronisation is by sequence numbers in device memory, no host round trip, no IPC events:
//   flags[src]  (one uint32 per source rank, in the RECEIVER's memory): "src's shard of step seq has landed here".
//               Written by src with a 4-byte copy queued behind the data copy on the same stream (stream order = the
//               data is complete before the flag is).
//   tp_gather_sync(wait_seq, publish)   a ONE-WAVE kernel on the compute stream: lane p polls flags[p] (system-scope
//               relaxed loads, s_sleep between polls, bounded by a wall-clock timeout that raises status[0]) until every
//               source has reached wait_seq, then stores `publish` into a cell the flag copies of the next push read.
//               It runs between two forwards, stream-ordered — never beside the persistent GEMMs.
// Back-pressure needs no acknowledgement traffic: with depth >= 3 and the contract "a step's tokens are consumed before
// the second-next submit", seeing a peer's flag of step i-1 proves that peer has passed its own sync of step i-1, i.e.
// has finished with the buffer of step i-3 = the buffer step i is about to overwrite (shard.DirectGather holds the proof).
#include "tp_internal.h"
#include <cstring>

namespace tp {

__global__ void __launch_bounds__(64)
gather_sync_kernel(const uint32_t* __restrict__ flags, const int world, const int rank, const uint32_t wait_seq,
                   uint32_t* __restrict__ seq_cell, const uint32_t publish_seq, int* __restrict__ status,
                   const long long timeout_ticks) {
    const int p = threadIdx.x;
    if (p < world && p != rank) {
        const long long t0 = wall_clock64();                       // 100 MHz, constant
        // (sequence numbers wrap: compare as a signed distance)
        while ((int32_t)(__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - wait_seq) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > timeout_ticks) {
                if (status) __hip_atomic_fetch_or(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    if (p == 0 && seq_cell) __hip_atomic_store(seq_cell, publish_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace tp

using namespace tp;

extern "C" {

int tp_gather_export(const void* ptr, void* handle, uint64_t* offset) {
    if (!ptr || !handle || !offset) { set_error("tp_gather_export: NULL argument"); return TP_ERR_INVALID_ARG; }
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess) { set_error("tp_gather_export: hipMemGetAddressRange: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    static_assert(sizeof(hipIpcMemHandle_t) == TP_IPC_HANDLE_BYTES, "handle size of the ABI");
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, (void*)base);
    if (e != hipSuccess) { set_error("tp_gather_export: hipIpcGetMemHandle: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    memcpy(handle, &h, sizeof(h));
    *offset = (uint64_t)((const char*)ptr - (const char*)base);
    return TP_OK;
}

int tp_gather_open(const void* handle, void** base) {
    if (!handle || !base) { set_error("tp_gather_open: NULL argument"); return TP_ERR_INVALID_ARG; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { set_error("tp_gather_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    *base = p;
    return TP_OK;
}

int tp_gather_close(void* base) {
    if (!base) { set_error("tp_gather_close: NULL argument"); return TP_ERR_INVALID_ARG; }
    const hipError_t e = hipIpcCloseMemHandle(base);
    if (e != hipSuccess) { set_error("tp_gather_close: hipIpcCloseMemHandle: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    return TP_OK;
}

// The flag words are written by OTHER devices' copy engines while a kernel of this device polls them.  Ordinary (coarse-grained)
// device memory is only guaranteed coherent with other agents at kernel boundaries — a polled line may sit in this device's L2.
// Fine-grained memory (what RCCL keeps its own flags in) is cached so that every agent sees every agent's writes.  torch's
// allocator cannot provide it, so this is the ONE allocation the library performs, on request: a few hundred bytes per gather.
int tp_gather_alloc_flags(void** flags, size_t bytes) {
    if (!flags || bytes == 0 || bytes > (1u << 20)) { set_error("tp_gather_alloc_flags: bad argument"); return TP_ERR_INVALID_ARG; }
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { set_error("tp_gather_alloc_flags: hipExtMallocWithFlags(fine-grained): %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();       // zeros in place before the handle leaves this process (set-up, not the data path)
    if (e != hipSuccess) { (void)hipFree(p); set_error("tp_gather_alloc_flags: hipMemset: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    *flags = p;
    return TP_OK;
}

int tp_gather_free_flags(void* flags) {
    if (!flags) { set_error("tp_gather_free_flags: NULL argument"); return TP_ERR_INVALID_ARG; }
    const hipError_t e = hipFree(flags);
    if (e != hipSuccess) { set_error("tp_gather_free_flags: hipFree: %s", hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    return TP_OK;
}

int tp_gather_sync(const uint32_t* flags, int world, int rank, uint32_t wait_seq, uint32_t* seq_cell, uint32_t publish_seq,
                   int32_t* status, int timeout_ms, void* stream) {
    if (!flags || world < 1 || world > 64 || rank < 0 || rank >= world) {
        set_error("tp_gather_sync: flags NULL or world / rank out of range (1 <= world <= 64)");
        return TP_ERR_INVALID_ARG;
    }
    const long long ticks = (long long)(timeout_ms > 0 ? timeout_ms : 30000) * 100000ll;      // wall_clock64: 100 MHz
    hipLaunchKernelGGL(gather_sync_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flags, world, rank, wait_seq, seq_cell,
                       publish_seq, (int*)status, ticks);
    return check_launch("gather_sync_kernel");
}

int tp_gather_push(int n_peers, void* const* dst, const void* src, size_t bytes, void* const* dst_flag, const uint32_t* seq_cell,
                   void* const* streams, int use_cus) {
    if (n_peers < 0 || (n_peers > 0 && (!dst || !dst_flag || !streams)) || !seq_cell || (bytes > 0 && !src)) {
        set_error("tp_gather_push: NULL argument");
        return TP_ERR_INVALID_ARG;
    }
    const hipMemcpyKind kind = use_cus ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToDeviceNoCU;
    for (int j = 0; j < n_peers; ++j) {
        if (!dst[j] || !dst_flag[j]) { set_error("tp_gather_push: peer %d has a NULL destination", j); return TP_ERR_INVALID_ARG; }
        hipStream_t st = (hipStream_t)streams[j];
        hipError_t e = hipSuccess;
        if (bytes > 0) e = hipMemcpyAsync(dst[j], src, bytes, kind, st);
        // the flag travels behind the data on the same stream: it cannot land before the shard has
        if (e == hipSuccess) e = hipMemcpyAsync(dst_flag[j], seq_cell, sizeof(uint32_t), kind, st);
        if (e != hipSuccess) { set_error("tp_gather_push: hipMemcpyAsync to peer %d: %s", j, hipGetErrorString(e)); return TP_ERR_LAUNCH; }
    }
    return TP_OK;
}

}  // extern "C"
