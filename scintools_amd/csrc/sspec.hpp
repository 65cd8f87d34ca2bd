// sspec.hpp -- the two-trip secondary-spectrum path (sspec.hip) behind scint_sspec (fft.hip).
#pragma once
#include "common.hpp"

namespace scint {

// shapes the fast path takes: halve = 1 and both half lengths next_pow2(nf), next_pow2(nt) in [256, 8192]
bool sspec_fast_supported(int64_t nf, int64_t nt, int32_t halve);
// bytes of its buffers: the intermediate Y[R/2][nt rounded up to even] (complex), the pair-major input copy, partial sums
size_t sspec_fast_workspace(int64_t nf, int64_t nt);
// windows / post-darkening tables as in scint_sspec; the means are computed inside (pass 0)
int32_t sspec_fast(const double* dyn, int64_t nf, int64_t nt, const double* win_t, const double* win_f,
                   int32_t prewhite, const double* pd_fd, const double* pd_td,
                   double* sec_out, void* workspace, hipStream_t stream);

}  // namespace scint
