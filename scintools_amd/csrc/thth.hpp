// thth.hpp -- CS <-> theta-theta maps (ththmod.py:56-271) as gfx950 kernels.
#pragma once
#include <float.h>
#include <math.h>

#include "common.hpp"

namespace scint {

// Device-side copy of the CS geometry plus the derived constants the reference
// computes once per call (ththmod.py:90-91): all host-computed with NumPy so the
// kernels see bit-identical operands.
struct GeomDev {
    int64_t ntau, nfd;
    double tau0, dtau, half_dtau;  // tau[0], diff(tau).mean(), dtau/2
    double fd0, dfd, half_dfd;     // fd[0],  diff(fd).mean(),  dfd/2
    double tau1_step, fd1_step;    // tau[1]-tau[0], fd[1]-fd[0]   (rev_map)
    double inv_dtau, inv_dfd;      // 1/dtau, 1/dfd: first guess of the floor (floor_div_exact_rcp)
    double ntau_d, nfd_d;          // the sizes as doubles (range tests on the un-converted quotients)
};
inline GeomDev to_dev(const scint_cs_geom& g) {
    GeomDev d;
    d.ntau = g.ntau; d.nfd = g.nfd;
    d.tau0 = g.tau0; d.dtau = g.dtau; d.half_dtau = g.dtau / 2;
    d.fd0 = g.fd0; d.dfd = g.dfd; d.half_dfd = g.dfd / 2;
    d.tau1_step = g.tau1_step; d.fd1_step = g.fd1_step;
    d.inv_dtau = 1.0 / g.dtau; d.inv_dfd = 1.0 / g.dfd;
    d.ntau_d = (double)g.ntau; d.nfd_d = (double)g.nfd;
    return d;
}

// ---- element math shared by the gather kernels ---------------------------------------
// (the translation units that CALL these are compiled with -ffp-contract=off)
// np.floor_divide(a, b) for float64, b > 0.  NumPy (npy_divmod) returns the exact
// mathematical floor(a/b): fmod is exact and the quotient is snapped to an integer.
// floor(fl(a/b)) can only be wrong (one too high) when the correctly-rounded quotient
// landed on an integer from below; the sign of the single-rounded remainder a - q*b
// detects exactly that case.
__device__ inline double floor_div_exact(double a, double b) {
    double q = floor(a / b);
    if (__builtin_fma(-q, b, a) < 0.0) q -= 1.0;
    return q;
}

// The same exact floor(a/b), b > 0, without the division: the first guess floor(a * (1/b)) is
// within one of the answer (|a/b| < 2^31 here, the product is good to ~2 ulp), and the sign of a
// single-rounded fma remainder is the sign of the exact remainder, so two sign tests settle it:
//   a - q b < 0        -> q was one too high
//   a - (q+1) b >= 0   -> q was one too low
// (testing "a - q b >= b" instead would be wrong: that remainder may round up onto b.)
__device__ inline double floor_div_exact_rcp(double a, double b, double inv_b) {
    const double q = floor(a * inv_b);
    const double r0 = __builtin_fma(-q, b, a);             // both remainders unconditionally, then pure
    const double r1 = __builtin_fma(-(q + 1.0), b, a);     // arithmetic (the two cases exclude each other)
    return (q - (r0 < 0.0 ? 1.0 : 0.0)) + (r1 >= 0.0 ? 1.0 : 0.0);
}

// np.nan_to_num on one float64: NaN -> 0, +-inf -> +-DBL_MAX.  Branch-free (two clamps and a select: as three `if`s the
// compiler built an exec-masked region per test -- six of them per element of the packed gather, round 4)
__device__ inline double nan_to_num(double v) {
    const double c = fmin(fmax(v, -DBL_MAX), DBL_MAX);      // (fmax / fmin return the other operand for a NaN)
    return v != v ? 0.0 : c;
}

// theta-theta value at (theta2 = th_i, theta1 = th_j), before any Hermitian forcing
// (ththmod.py:94-107).
__device__ inline cplx thth_value(const cplx* __restrict__ cs, const GeomDev& g, double eta,
                                  double two_eta, double th_i, double th_j) {
    const double a_tau = ((eta * (th_j * th_j - th_i * th_i)) - g.tau0) + g.half_dtau;
    const double a_fd = ((th_j - th_i) - g.fd0) + g.half_dfd;
    const int64_t tau_inv = (int64_t)floor_div_exact(a_tau, g.dtau);
    int64_t fd_inv = (int64_t)floor_div_exact(a_fd, g.dfd);
    // pnts = (tau_inv > 0) * (tau_inv < ntau) * (fd_inv < nfd): no lower bound on
    // fd_inv (ththmod.py:103); NumPy's fancy index wraps a negative one.
    if (!(tau_inv > 0 && tau_inv < g.ntau && fd_inv < g.nfd)) return mk(0.0, 0.0);
    if (fd_inv < 0) {
        fd_inv += g.nfd;
        if (fd_inv < 0) return mk(nan(""), nan(""));  // NumPy would raise IndexError
    }
    const cplx v = cs[tau_inv * g.nfd + fd_inv];
    const double w = sqrt(fabs(two_eta * (th_i - th_j)));
    return mk(v.x * w, v.y * w);
}

// Index half of thth_value: linear offset into CS of the pixel feeding (theta2 = th_i,
// theta1 = th_j), -1 if the point is outside the CS (value 0), -2 if NumPy would raise.
__device__ inline int64_t thth_offset(const GeomDev& g, double eta, double th_i, double th_j) {
    const double a_tau = ((eta * (th_j * th_j - th_i * th_i)) - g.tau0) + g.half_dtau;
    const double a_fd = ((th_j - th_i) - g.fd0) + g.half_dfd;
    const int64_t tau_inv = (int64_t)floor_div_exact(a_tau, g.dtau);
    int64_t fd_inv = (int64_t)floor_div_exact(a_fd, g.dfd);
    if (!(tau_inv > 0 && tau_inv < g.ntau && fd_inv < g.nfd)) return -1;
    if (fd_inv < 0) {
        fd_inv += g.nfd;
        if (fd_inv < 0) return -2;
    }
    return tau_inv * g.nfd + fd_inv;
}

// One theta-theta matrix to build: curvature, crop and destination.
struct GatherJob {
    double eta, two_eta;     // eta and 2*eta (ththmod.py:95, 107)
    const int32_t* keep;     // N indices into th_cents (ascending)
    int32_t n;               // N
    int32_t hermitian;
    cplx* out;               // [N, ld]
    int64_t ld;
};

// Enqueue the full-matrix gather of one job on `stream` (asynchronous).
int32_t launch_gather(const cplx* cs, const GeomDev& g, const double* th_cents, int64_t M,
                      const GatherJob& job, hipStream_t stream);

// Enqueue the rank-1 back-map  recov = rev_map(|w| v v^H)  (modeler, ththmod.py:312-321) on
// `stream`: th[N] are the centres of the reduced edges, w a DEVICE scalar, scratch 256 bytes.
// `transposed`: write recov^T [nfd, ntau] instead of recov [ntau, nfd].
int32_t launch_rev_map_rank1(const cplx* vec, const double* w, const double* th, int64_t N, const GeomDev& g,
                             double eta, cplx* recov, bool transposed, void* scratch, hipStream_t stream);


// ---- the same back-map for SEVERAL retired curvatures in one set of launches (chi^2 sweep, round 5) -----------------------
// A chunk of the sweep retires its curvatures together (they start together and need about the same number of passes: 20-30 per
// chunk at 4096^2), and the per-curvature tail -- two memsets, bound, setup, back-map, chi^2, final sum, four profiling events --
// cost the sweep's ONE host thread 11 API calls each, during which neither slot group got its next chunk: the mat-vec was in
// flight for only 85 % of a chi^2 step (profiles/r04_timeline.txt; 97.5 % in the eigenvalue sweep).  Batched: four launches
// per <= kRevBatchMax curvatures, whatever their number.
#ifndef SCINT_REV_BATCH
#define SCINT_REV_BATCH 8
#endif
constexpr int kRevBatchMax = SCINT_REV_BATCH;
struct RevJobDev {                 // one per curvature of the sweep, built on the host before the sweep starts
    const cplx* vec; const double* w; const double* th;   // eigenvector [N], |w| (device scalar), centres of the reduced edges [N]
    double eta, two_eta, inv_tau1_step;
    int64_t centre;                // flat index of the pixel the i == j terms poison, or -1
    unsigned long long* bound;     // [kRevWords] this curvature's constants (rev_prep_batch_kernel writes, the gather and chi^2 read)
    const uint8_t* walk;           // [nfd][N] partner masks of the curvature's CROP (launch_rev_walk_table), or nullptr: walk in the kernel
    const uint8_t* walk_col;       // [nfd]    1 = the column's masks are complete (every lane's window brackets the column)
    int32_t N, pad;
};
constexpr int kRevWords = 16;      // words per curvature: max |v| sqrt 2, min spacing, the RevConsts, the delay band
struct RevBatch {                  // by value in the kernel arguments
    int32_t n, pad;
    int32_t job[kRevBatchMax];     // index into the RevJobDev table
    cplx* recov[kRevBatchMax];     // transposed image [nfd, ntau] of each
};
RevJobDev make_rev_job(const cplx* vec, const double* w, const double* th, int64_t N, const GeomDev& g, double eta,
                       unsigned long long* bound);
// bound pre-pass + constants + delay band (one workgroup per curvature), then the column gather over grid.z = curvature.
// Only the delay rows a curvature can reach -- |tau| <= |eta| max theta^2, widened to a band that is symmetric about
// tau = 0 -- are computed and written (whole slabs that miss the band leave at once); the band is left in
// bound[kRevBandLo], bound[kRevBandHi] for the consumer (chisq_parseval_batch_kernel), which must not read outside it.
// `uniform[job]` (host, or nullptr): the flags launch_rev_uniform left -- those curvatures take the diagonal kernel (thth.hip), the
// others the general one; a job whose device flag disagrees with the host's copy is formed by neither (the flags are one read-back
// of one kernel's output, so they agree).
// `fuse` (or nullptr): the uniform-grid curvatures of the batch do not write their image -- every workgroup leaves the chi^2 terms
// of its interior pixels (Doppler column >= 1, delay row >= 1, inside the band), sum |recov - spec|^2, in
// partial[image * partial_stride + column * slabs + slab] (image = position among the batch's uniform-grid curvatures; 0 for a
// slab outside the band), writes only Doppler column 0 and delay row 0 of recov^T (the pixels whose mirror pixel is not the
// mirror image: the consumer forms their terms), and raises asym[job] when a pair sat on a bin edge (the histogram may then
// not be mirror-symmetric: that curvature's chi^2 must be formed from a written image).  spec = fftshift(fft2(dspec^T)) [nfd][ntau].
// `general_out` / `uniform_out` (or nullptr): the two halves of the batch, in the order of their images.
struct RevFuse { const cplx* spec; double* partial; int64_t partial_stride; int32_t* asym; };
int32_t launch_rev_map_rank1_batch(const RevJobDev* jobs_dev, const RevBatch& b, const GeomDev& g, const uint8_t* uniform, const RevFuse* fuse,
                                   RevBatch* general_out, RevBatch* uniform_out, hipStream_t stream);
int64_t rev_diag_items(const GeomDev& g);      // work items (Doppler columns x delay slabs) of one image of the uniform-grid kernel
int64_t rev_diag_items_for(int64_t ntau, int64_t nfd);
// The grid test of every job of the table (rank-1 Hermitian back-map on a uniform theta grid): writes bound[kRevUniform],
// bound[kRevSlack] of each and flags[job] = 0 / 1.  Once per sweep, before launch_rev_map_rank1_batch.
int32_t launch_rev_uniform(const RevJobDev* jobs_dev, int64_t njobs, const GeomDev& g, int32_t* flags_dev, hipStream_t stream);
// Which theta_j are partners of theta_i in Doppler column c does not depend on the curvature -- fd_map = theta_j - theta_i
// (ththmod.py:207) -- only on the theta centres, i.e. on the CROP.  Curvatures that keep the same centres (161 of the 256 of
// the headline sweep keep all 4095) can therefore share the part of the back-map that finds those partners (40 % of the
// kernel): masks[c][i] has bit k - 1 set iff theta_{i + s0(c) - 1 + k} is a partner of theta_i in column c (the kernel's own
// window walk, k = 1 .. W <= 8), col_ok[c] = 1 iff every lane's window brackets the column (always on a near-uniform grid;
// otherwise the column keeps the in-kernel walk).  One launch per crop; `masks` nfd * N bytes, `col_ok` nfd bytes.
int32_t launch_rev_walk_table(const double* th, int64_t N, const GeomDev& g, uint8_t* masks, uint8_t* col_ok, hipStream_t stream);
enum { kRevS1 = 2, kRevS2 = 3, kRevExact = 4, kRevThStep = 5, kRevW = 6, kRevBandLo = 7, kRevBandHi = 8, kRevUniform = 9, kRevSlack = 10 };

// np.histogram2d bin of x on the edges e(k) = (k - 0.5)*step + x0, k = 0..n:
// searchsorted(edges, x, 'right') - 1, with x == e(n) folded into the last bin
// (numpy/lib/_histograms_impl.py histogramdd).  Returns -1 for an outlier.
__host__ __device__ inline int64_t hist_bin(double x, double x0, double step, int64_t n) {
    if (!(step > 0.0) || x != x) return -1;
    double guess = floor((x - x0) / step + 0.5);
    if (guess < -1.0) return -1;
    if (guess > (double)n + 1.0) return -1;
    int64_t k = (int64_t)guess;
    if (k < 0) k = 0;
    if (k > n) k = n;
    // largest k in [0, n] with e(k) <= x
    while (k < n && (((double)(k + 1) - 0.5) * step + x0) <= x) ++k;
    while (k >= 0 && (((double)k - 0.5) * step + x0) > x) --k;
    if (k < 0) return -1;
    if (k == n) return (x == (((double)n - 0.5) * step + x0)) ? n - 1 : -1;
    return k;
}


}  // namespace scint
