// capi.hip -- error plumbing and diagnostics of the C ABI (include/scint_hip.h).
#include <stdio.h>
#include <string.h>

#include "prof.hpp"

namespace scint {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int32_t hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "scint: HIP error %d (%s) in `%s` at %s:%d", (int)e,
             hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    return SCINT_E_HIP;
}

Profiler& profiler() {
    static Profiler p;
    return p;
}

}  // namespace scint

extern "C" int32_t scint_profile_begin(void) {
    scint::Profiler& p = scint::profiler();
    p.reset(nullptr);
    p.enabled = true;
    return SCINT_OK;
}

extern "C" int32_t scint_profile_end(double* ms_out, double* ms_sum_out, int64_t* launches_out, int32_t count) {
    scint::Profiler& p = scint::profiler();
    if (count < 0) { scint::set_error("scint: profile_end: negative count"); return SCINT_E_ARG; }
    if (hipDeviceSynchronize() != hipSuccess) return SCINT_E_HIP;
    p.collect();
    p.finish();
    p.enabled = false;
    for (int k = 0; k < scint::kProfCount && k < count; ++k) {       // never more than the caller's arrays hold
        if (ms_out) ms_out[k] = p.ms[k];
        if (ms_sum_out) ms_sum_out[k] = p.ms_sum[k];
        if (launches_out) launches_out[k] = p.launches[k];
    }
    return SCINT_OK;
}

extern "C" int32_t scint_version(void) { return 107; }   // 107: the rank-1 Hermitian back-map on a uniform theta grid is rev_diag_kernel (scint_rev_map leaves which kernel ran in word 9 of its workspace; SCINT_REV_DIAG=0 keeps the general kernel); 106: scint_retrieval_tail takes class_id (one-row back-map, pair counts per class); 105: scint_mosaic_* / scint_chunk_cut; 104: calc_sspec through persistent kernels (no signature change; the version binds library and package: _lib.load checks it); 101: scint_profile_end takes the length of the caller's arrays; 102: scint_sweep_workgroups; 103: scint_chisq_sweep takes crop_group

extern "C" int32_t scint_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return SCINT_E_ARG;
    strncpy(buf, scint::g_last_error.c_str(), n - 1);
    buf[n - 1] = '\0';
    return SCINT_OK;
}

extern "C" int32_t scint_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
