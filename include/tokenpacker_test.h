/* tokenpacker_test.h — test hooks and probe builds of the TokenPacker HIP library.  NOT part of the product ABI
 * (include/tokenpacker.h): these entry points exist only in libtokenpacker_exp.so (`make -C tokenpacker_amd/csrc exp`, built by
 * __graft_entry__.build() beside the product library from the same sources + csrc/tp_test_hooks.hip, the GEMM kernels' timing-probe
 * instantiations (-DTP_BUILD_PROBES: TP_TUNE_PAIR_DEBUG) and csrc/experimental/).  All of them are stateless: a test may call them
 * in libtokenpacker_exp.so while the module under test runs on libtokenpacker_hip.so. */
#ifndef TOKENPACKER_TEST_H
#define TOKENPACKER_TEST_H
#include "tokenpacker.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- test hook: occupy `workgroups` CUs for ~`microseconds` on `stream` (100 KiB LDS each; `scratch_int`: any
 * device int).  Stands in for another stream's kernels when the GEMM tile queue is measured (tools/hog_bench.py). */
int tp_test_occupy_cus(int workgroups, int microseconds, void* scratch_int, void* stream);
/* ---- test hook: workgroups of the pair kernel the runtime admits per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor;
 * the design needs 2); negative on error */
int tp_test_pair_occupancy(void);
/* ---- test hook (host logic only, no GPU needed): where tp_linear would send a plain fp16 launch of this shape under the tuning
 * of the moment — 0 the 128-tile kernel | 1 full 256 x 256 tiles | 2 all 128 x 256 half tiles | 3 full rounds + a half-tile tail
 * launch | 4 192 x 256 tiles | 5 the pair kernel; -1: bad shape.  (256 CUs are assumed when no device is visible.) */
int tp_test_gemm_route(int M, int N, int K, int flags, int groups);
/* ---- test hook: the pack-time factorisation behind TP_TUNE_TRI_STATS on ONE layer.  w2 [1024][1024] fp16 and b2 [1024] fp32
 * (or NULL) in; r [1024][1024] fp16 (upper triangular), c_tilde [1024] fp32 and wbar [1025] fp32 (column means of w2, then
 * mean(b2)) out, with  sum_n ((w2 h + b2)_n - mean)^2 = || r h + c_tilde ||^2  for every h.  scratch: device memory of
 * tp_test_pack_qr_scratch_bytes() bytes. */
size_t tp_test_pack_qr_scratch_bytes(void);
int tp_test_pack_qr(const void* w2_f16, const float* b2, void* r_f16, float* c_tilde, float* wbar, void* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif
