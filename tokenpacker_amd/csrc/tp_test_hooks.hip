// tp_test_hooks.hip — the test hooks of include/tokenpacker_test.h.  Linked into libtokenpacker_exp.so only (Makefile target `exp`);
// the product library does not export them.  Stateless: each reads nothing but its arguments and the device.
#include "tp_internal.h"
#include "../../include/tokenpacker_test.h"

using namespace tp;
#define TP_TRY(expr) do { int rc_ = (expr); if (rc_ != TP_OK) return rc_; } while (0)

extern "C" {

int tp_test_occupy_cus(int workgroups, int microseconds, void* scratch_int, void* stream) {
    if (workgroups <= 0 || microseconds <= 0 || !scratch_int) { set_error("tp_test_occupy_cus: bad argument"); return TP_ERR_INVALID_ARG; }
    return occupy_cus_launch(workgroups, microseconds, (int*)scratch_int, (hipStream_t)stream);
}


int tp_test_pair_occupancy(void) { return gemm_pair_occupancy(); }
int tp_test_gemm_route(int M, int N, int K, int flags, int groups) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 128 != 0 || K % 64 != 0) { set_error("tp_test_gemm_route: bad shape"); return -1; }
    GemmArgs a = plain_gemm(nullptr, K, nullptr, nullptr, N, M, N, K, nullptr, flags);
    a.groups = groups > 0 ? groups : 1;
    return gemm_route_of(TP_F16, TP_F16, a);
}

size_t tp_test_pack_qr_scratch_bytes(void) { return pack_qr_scratch_bytes(1); }

int tp_test_pack_qr(const void* w2_f16, const float* b2, void* r_f16, float* c_tilde, float* wbar, void* scratch, void* stream) {
    if (!w2_f16 || !r_f16 || !c_tilde || !wbar || !scratch) { set_error("tp_test_pack_qr: NULL argument"); return TP_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    TP_TRY(pack_qr_center_launch(w2_f16, b2, scratch, 0, wbar, st));
    TP_TRY(pack_qr_factor_launch(scratch, 1, st));
    return pack_qr_extract_launch(scratch, 0, r_f16, c_tilde, st, nullptr);
}


}  // extern "C"
