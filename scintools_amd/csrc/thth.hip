// thth.hip -- CS -> theta-theta gather (thth_map + thth_redmap, ththmod.py:56-173)
// and theta-theta -> CS scatter (rev_map, ththmod.py:176-271).
//
// This translation unit is compiled with -ffp-contract=off: the bin an element
// falls in is decided by a floor / an ordered comparison of float64 expressions,
// and the reference evaluates those expressions with one IEEE rounding per NumPy
// ufunc.  A fused multiply-add would flip pixels.
#include <map>
#include <type_traits>
#include <float.h>
#include <limits.h>
#include <math.h>

#include "packed.hpp"
#include "prof.hpp"
#include "thth.hpp"

namespace scint {

constexpr int kTile = 32;

// One 32x32 tile of one job per 256-thread block.  Hermitian jobs compute only tiles on
// or above the diagonal; the mirrored tile is written through an LDS transpose so that
// both the (i, j) and the (j, i) stores are coalesced row segments.
__global__ void __launch_bounds__(256)
thth_gather_kernel(const cplx* __restrict__ cs, GeomDev g, const double* __restrict__ th,
                   int64_t M, GatherJob job) {
    __shared__ cplx tile[kTile][kTile + 1];
    const int N = job.n;
    const int I0 = blockIdx.y * kTile, J0 = blockIdx.x * kTile;
    if (I0 >= N || J0 >= N) return;
    if (job.hermitian && J0 < I0) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int j = J0 + tx;
    const int kj = j < N ? job.keep[j] : 0;
    const double th_j = j < N ? th[kj] : 0.0;

    if (!job.hermitian) {
        for (int r = ty; r < kTile; r += 8) {
            const int i = I0 + r;
            if (i < N && j < N)
                job.out[(int64_t)i * job.ld + j] =
                    thth_value(cs, g, job.eta, job.two_eta, th[job.keep[i]], th_j);
        }
        return;
    }

    const bool diag_tile = (I0 == J0);
    for (int r = ty; r < kTile; r += 8) {
        const int i = I0 + r;
        cplx v = mk(0.0, 0.0);
        if (i < N && j < N && i < j) {
            const int ki = job.keep[i];
            v = thth_value(cs, g, job.eta, job.two_eta, th[ki], th_j);
            // anti-diagonal of the FULL matrix is zeroed (ththmod.py:113), then nan_to_num
            if ((int64_t)ki + kj == M - 1) v = mk(0.0, 0.0);
            v = mk(nan_to_num(v.x), nan_to_num(v.y));
        }
        tile[r][tx] = v;
        if (!diag_tile && i < N && j < N) job.out[(int64_t)i * job.ld + j] = v;
    }
    __syncthreads();
    if (diag_tile) {
        for (int r = ty; r < kTile; r += 8) {
            const int i = I0 + r;
            if (i < N && j < N) {
                cplx v = (r < tx) ? tile[r][tx] : (r > tx ? conj(tile[tx][r]) : mk(0.0, 0.0));
                job.out[(int64_t)i * job.ld + j] = v;
            }
        }
    } else {
        // mirrored tile: out[J0 + a][I0 + b] = conj(tile[b][a]), b fastest
        for (int a = ty; a < kTile; a += 8) {
            const int row = J0 + a, col = I0 + tx;
            if (row < N && col < N) job.out[(int64_t)row * job.ld + col] = conj(tile[tx][a]);
        }
    }
}

int32_t launch_gather(const cplx* cs, const GeomDev& g, const double* th_cents, int64_t M,
                      const GatherJob& job, hipStream_t stream) {
    if (job.n <= 0) return SCINT_OK;
    const unsigned nt = (unsigned)ceil_div(job.n, kTile);
    SCINT_REQUIRE(nt <= 65535, "gather: grid too large");
    const int slot = profiler().begin(kProfGather, stream);
    hipLaunchKernelGGL(thth_gather_kernel, dim3(nt, nt, 1), dim3(256), 0, stream, cs, g, th_cents, M, job);
    profiler().end(kProfGather, slot, stream);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// packed gather (eta sweep): only tiles on/above the block diagonal, 64 KiB contiguous each
// ------------------------------------------------------------------------------
// One 64 x 64 packed tile per 256-thread workgroup.  Lanes run along the 64 columns (every row a
// wave stores is one contiguous 1 KiB segment of the tile), wave w owns rows 16w .. 16w+15, i.e.
// 16 elements per lane, handled as two batches of 8 whose scattered 16-B CS reads are all issued
// before the first result of the batch is needed.  Everything that
// depends on the row only (keep index, theta_i, theta_i^2) is wave-uniform and is read through
// the scalar unit; the column's theta_j costs each lane ONE dependent load pair for the whole
// tile.  The bin index is the exact floor of thth.hpp, computed without a division.
//
// Non-finite geometry (a_tau or a_fd NaN/inf) gives q = NaN -> the range test fails -> 0, which
// is what the reference's  pnts  mask does with the int-converted NaN as well.
struct GatherElem { int32_t off; double wgt; };   // off: element index into the conjugate spectrum (launch_gather_packed
                                                  // refuses spectra of 2^31 elements or more), -1 = zero, -2 = NaN marker

// Branch-free: `live` false (outside the matrix, the diagonal, the anti-diagonal) gives off = -1.
__device__ inline GatherElem gather_elem(const GeomDev& g, int nfd, double eta, double two_eta, double t2, double t1,
                                         double sq2, double sq1, bool live) {
    // (theta2 = t2, theta1 = t1): ththmod.py:94-97 in the reference's operation order
    const double a_tau = ((eta * (sq1 - sq2)) - g.tau0) + g.half_dtau;
    const double a_fd = ((t1 - t2) - g.fd0) + g.half_dfd;
    const double qt = floor_div_exact_rcp(a_tau, g.dtau, g.inv_dtau);
    const double qf = floor_div_exact_rcp(a_fd, g.dfd, g.inv_dfd);
    // pnts = (tau_inv > 0) * (tau_inv < ntau) * (fd_inv < nfd)   (ththmod.py:103); comparisons
    // with NaN are false.  Inside that range both quotients fit an int32 except a very negative
    // fd index, which NumPy would reject (IndexError): -2.
    const bool in = live && qt > 0.0 && qt < g.ntau_d && qf < g.nfd_d;
    const bool wrap_ok = qf >= -g.nfd_d;
    const int it = __double2int_rz(qt);                 // saturating conversions: harmless when !in
    int jf = __double2int_rz(qf);
    jf += jf < 0 ? nfd : 0;                             // NumPy's negative-index wrap
    GatherElem e;
    // (in range: it < ntau, 0 <= jf < nfd, ntau nfd < 2^31.  Out of range `it` is saturated and the product would overflow a
    //  signed int -- undefined, though the select discards it -- so the index is formed in unsigned arithmetic: ADVICE r4)
    const int idx = (int)((unsigned)it * (unsigned)nfd + (unsigned)jf);
    e.off = in ? (wrap_ok ? idx : -2) : -1;
    e.wgt = sqrt(fabs(two_eta * (t2 - t1)));
    return e;
}

// packed tile index -> (I, J): largest I with tile_offset(nb, I) <= t
__device__ inline void tile_coords(int nb, int t, int& I, int& J) {
    const float s = 2.0f * (float)nb + 1.0f;
    int i = (int)((s - sqrtf(fmaxf(s * s - 8.0f * (float)t, 0.0f))) * 0.5f);
    i = min(max(i, 0), nb - 1);
    while (i > 0 && tile_offset(nb, i) > t) --i;
    while (i + 1 < nb && tile_offset(nb, i + 1) <= t) ++i;
    I = i;
    J = i + (t - (int)tile_offset(nb, i));
}

// F32: the tile also leaves as complex64, multiplied by the job's power-of-two scale (exact) and rounded once
// -- the operand of the iteration phase of the mixed-precision sweep (eigen_packed.hip).
template <bool F32>
__global__ void __launch_bounds__(256)
thth_gather_packed_kernel(const GeomDev* __restrict__ geoms, int64_t M,
                          const PackedJob* __restrict__ jobs, const int32_t* __restrict__ slots) {
    const PackedJob* __restrict__ jp = jobs + slots[blockIdx.y];
    const int nb = jp->nb, n = jp->n;
    const int t = blockIdx.x;
    if (t >= tile_count(nb)) return;
    int I, J;
    tile_coords(nb, t, I, J);
    const cplx* __restrict__ cs = jp->cs;
    const double* __restrict__ th = jp->th;
    const int32_t* __restrict__ keep = jp->keep;
    const GeomDev g = geoms[jp->geom];
    const double eta = jp->eta, two_eta = jp->two_eta;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform row group
    const int j = J * kTB + lane;
    const bool jin = j < n;
    const int kj = jin ? gload(keep + j) : 0;
    const double th_j = jin ? gload(th + kj) : 0.0;
    const double sq_j = th_j * th_j;
    cplx* __restrict__ tile = jp->tiles + (int64_t)t * kTileElems + lane;
    c32* __restrict__ tile32 = F32 ? jp->tiles32 + (int64_t)t * kTileElems + lane : nullptr;
    const double sc32 = F32 ? gload(jp->scale32) : 1.0;
    const bool diag = (I == J);
    // the wave's 16 rows: lanes 0..15 fetch keep / theta_i once (one dependent load pair), every
    // row then reads them back through the scalar unit
    const int i_l = I * kTB + 16 * w + (lane & 15);
    const int ki_l = i_l < n ? gload(keep + i_l) : 0;
    const double thi_l = i_l < n ? gload(th + ki_l) : 0.0;
    const int nfd = (int)g.nfd;
    const int Mm1 = (int)(M - 1);
    GatherElem el[2][8];
    cplx val[2][8];
    // DIAG is uniform over the workgroup (the tile's coordinates): an off-diagonal tile (2016 of the 2080 at N = 4095) has
    // i < j for every element, so the selects between "upper element" and "conjugate of the mirrored one" -- four
    // 64-bit selects and a sign per element -- exist only in the copy of the loop the 64 diagonal tiles run (round 4:
    // 126 vector instructions per element before, profiles/r03_sweep_counters.txt)
    auto index_and_load = [&](int b, auto diag_c) {
        constexpr bool DIAG = decltype(diag_c)::value;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = I * kTB + 16 * w + 8 * b + k;    // wave-uniform
            const int ki = __builtin_amdgcn_readlane(ki_l, 8 * b + k);
            const double th_i = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(thi_l), 8 * b + k),
                                                 __builtin_amdgcn_readlane(__double2loint(thi_l), 8 * b + k));
            const double sq_i = th_i * th_i;
            const bool anti = ki + kj == Mm1;              // anti-diagonal: ththmod.py:113
            if (DIAG) {
                // upper element (i < j) reads (theta2 = th_i, theta1 = th_j); the lower half of a
                // diagonal tile is the conjugate of the mirrored upper element
                const bool up = i < j;
                const bool live = i < n && jin && i != j && !anti;
                el[b][k] = gather_elem(g, nfd, eta, two_eta, up ? th_i : th_j, up ? th_j : th_i, up ? sq_i : sq_j,
                                       up ? sq_j : sq_i, live);
                if (!up) el[b][k].wgt = -el[b][k].wgt;     // sign carries "conjugate"
            } else {
                el[b][k] = gather_elem(g, nfd, eta, two_eta, th_i, th_j, sq_i, sq_j, i < n && jin && !anti);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) val[b][k] = gload(cs + (el[b][k].off >= 0 ? el[b][k].off : 0));
    };
    auto weight_and_store = [&](int b, auto diag_c) {
        constexpr bool DIAG = decltype(diag_c)::value;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const GatherElem e = el[b][k];
            const double aw = fabs(e.wgt);
            double vx = nan_to_num(val[b][k].x * aw), vy = nan_to_num(val[b][k].y * aw);
            if (DIAG) vy = e.wgt < 0.0 ? -vy : vy;
            // off == -1: outside the CS / diagonal / anti-diagonal -> 0;  off == -2: NumPy would
            // raise IndexError (cannot happen in a Hermitian job) -> NaN marker
            const double none = e.off == -2 ? nan("") : 0.0;
            const cplx v = mk(e.off >= 0 ? vx : none, e.off >= 0 ? vy : none);
            gstore_nt(tile + (16 * w + 8 * b + k) * kTB, v);   // written once, read much later
            if (F32) gstore_nt(tile32 + (16 * w + 8 * b + k) * kTB, v.x * sc32, v.y * sc32);
        }
    };
    // batch by batch (8 reads in flight per lane, ~100 VGPRs, 4 waves per SIMD); issuing both
    // batches' reads first (168 VGPRs, 3 waves) measured the same
    if (diag) {
        const std::true_type d;
        index_and_load(0, d); weight_and_store(0, d); index_and_load(1, d); weight_and_store(1, d);
    } else {
        const std::false_type d;
        index_and_load(0, d); weight_and_store(0, d); index_and_load(1, d); weight_and_store(1, d);
    }
}

int32_t launch_gather_packed(const GeomDev* geoms_dev, int64_t M, const PackedJob* jobs_dev,
                             const int32_t* slots_dev, int njobs, int nbmax, hipStream_t stream, bool with32) {
    if (njobs <= 0 || nbmax <= 0) return SCINT_OK;
    SCINT_REQUIRE(nbmax <= 32767 && njobs <= 65535, "gather: grid too large");
    SCINT_REQUIRE(M <= INT32_MAX, "gather: theta grid too large");
    const int slot = profiler().begin(kProfGather, stream);
    const dim3 grid((unsigned)tile_count(nbmax), (unsigned)njobs);
    if (with32) hipLaunchKernelGGL(thth_gather_packed_kernel<true>, grid, dim3(256), 0, stream, geoms_dev, M, jobs_dev, slots_dev);
    else hipLaunchKernelGGL(thth_gather_packed_kernel<false>, grid, dim3(256), 0, stream, geoms_dev, M, jobs_dev, slots_dev);
    profiler().end(kProfGather, slot, stream);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// max(|re|, |im|) over the finite elements of a conjugate spectrum, as the bit pattern of a non-negative double
// (integer order = numeric order: atomicMax is exact and order-independent), then the power of two that brings
// it into [0.5, 1).  The complex64 copy of theta-theta is gathered from the spectrum times this scale, so the
// iteration phase of the mixed sweep does not depend on the units of the dynamic spectrum.
__global__ void __launch_bounds__(256) cs_absmax_kernel(const cplx* __restrict__ cs, int64_t cs_stride, int64_t nelem,
                                                        unsigned long long* __restrict__ bits) {
    __shared__ double red[4];
    const cplx* __restrict__ src = cs + (int64_t)blockIdx.y * cs_stride;
    double m = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nelem; k += (int64_t)gridDim.x * 256) {
        const cplx v = gload(src + k);
        const double a = fabs(v.x), b = fabs(v.y);
        if (a <= 1.7976931348623157e308) m = fmax(m, a);       // NaN and inf fail the comparison
        if (b <= 1.7976931348623157e308) m = fmax(m, b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(bits + blockIdx.y, (unsigned long long)__double_as_longlong(m));
    }
}
__global__ void __launch_bounds__(64) cs_scale_kernel(const unsigned long long* __restrict__ bits, double* __restrict__ scale,
                                                      int64_t ncs) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= ncs) return;
    const double m = __longlong_as_double((long long)bits[c]);
    int e = 0;
    if (m > 0.0) (void)frexp(m, &e);                           // m = f 2^e, f in [0.5, 1)
    e = e > 1000 ? 1000 : (e < -1000 ? -1000 : e);
    scale[c] = m > 0.0 ? ldexp(1.0, -e) : 1.0;
}

int32_t launch_cs_scale(const cplx* cs, int64_t ncs, int64_t cs_stride, int64_t nelem, unsigned long long* bits,
                        double* scale, hipStream_t stream) {
    SCINT_REQUIRE(ncs >= 1 && nelem >= 1, "cs_scale: bad arguments");
    SCINT_HIP(hipMemsetAsync(bits, 0, sizeof(unsigned long long) * (size_t)ncs, stream));
    const unsigned nblk = (unsigned)std::min<int64_t>(ceil_div(nelem, 256 * 8), 2048);
    for (int64_t c0 = 0; c0 < ncs; c0 += 65535) {               // (grid.y is limited to 65535)
        const unsigned nc = (unsigned)std::min<int64_t>(ncs - c0, 65535);
        hipLaunchKernelGGL(cs_absmax_kernel, dim3(nblk, nc), dim3(256), 0, stream, cs + c0 * cs_stride, cs_stride, nelem, bits + c0);
    }
    hipLaunchKernelGGL(cs_scale_kernel, dim3((unsigned)ceil_div(ncs, 64)), dim3(64), 0, stream, bits, scale, ncs);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// rev_map
// ------------------------------------------------------------------------------
// (hist_bin: thth.hpp)

// The same bin for the back-map kernel, 32-bit index, the first guess from a multiplication by 1/step.  The
// edges e(k) = (k - 0.5) step + x0 are non-decreasing in k, so the answer is the largest k in [0, n) with
// e(k) <= x < e(k + 1); a guess that is off by at most one (any sane x: the guess only carries the rounding of
// one subtraction and one multiplication) is settled with two or three edge evaluations and no loop.  Whatever
// that does not prove -- a guess further off, the closed last bin, an x outside the axis -- takes the loops of
// hist_bin.  (The loops alone cost every pair sixty instructions: the compiler unrolls the second one four-fold,
// `#pragma clang loop unroll(disable)` notwithstanding.)
__device__ inline int hist_bin_rcp(double x, double x0, double step, double inv_step, int n) {
    if (!(step > 0.0) || x != x) return -1;
    const double guess = floor((x - x0) * inv_step + 0.5);
    if (guess < -1.0) return -1;
    if (guess > (double)n + 1.0) return -1;
    int k = (int)guess;
    if (k < 0) k = 0;
    if (k > n) k = n;
    auto edge = [&](int q) { return ((double)q - 0.5) * step + x0; };
    const double ea = edge(k), eb = edge(k + 1);
    bool sure;
    if (k < n && eb <= x) { ++k; sure = k < n && edge(k + 1) > x; }
    else if (ea > x) { --k; sure = k >= 0 && edge(k) <= x; }
    else sure = k < n;
    if (sure) return k;
    while (k < n && edge(k + 1) <= x) ++k;
    while (k >= 0 && edge(k) > x) --k;
    if (k < 0) return -1;
    if (k == n) return (x == edge(n)) ? n - 1 : -1;
    return k;
}

struct RevParams {
    const cplx* thth; int64_t ld;      // explicit matrix (rank1 == 0)
    const cplx* vec; const double* w;  // rank-1: |w| v v^H (rank1 == 1)
    int rank1;
    const double* th; int N;
    double eta, two_eta;
    int hermitian;
    int slab;         // tau rows per workgroup
    int64_t centre;   // flat index of the pixel the i == j terms poison, or -1
    cplx* recov;      // [ntau, nfd], or its transpose [nfd, ntau] when `transposed`
    int transposed;   // column-major output: every workgroup writes one contiguous run (chi^2 sweep)
    const unsigned long long* bound;   // [2] bit patterns: max |value|, min theta spacing (rev_bound_kernel);
                                       // [2..6] the per-image constants of rev_setup_kernel (RevConsts)
    double inv_tau1_step;              // 1 / tau1_step (host: the same IEEE quotient the kernel used to form per wavefront)
    const uint8_t* walk;               // partner masks of the crop (thth.hpp: launch_rev_walk_table), or nullptr
    const uint8_t* walk_col;
    int stride_max;                    // rev_diag_body: strided sweeps up to this many passes, the segmented scan beyond
    // rev_diag_body<FUSE>: chi^2 straight from the accumulators (RevFuse of thth.hpp); the image is not written
    const cplx* spec; double* partial; int32_t* asym; int band_lo, band_hi;
};

#ifndef SCINT_REV_SLAB
#define SCINT_REV_SLAB 1024        // (build constants of the round-5 A/Bs: tools/build_variant.sh -DSCINT_REV_SLAB=320 -DSCINT_REV_CAP=512 ...)
#endif
#ifndef SCINT_REV_CAP
#define SCINT_REV_CAP 0
#endif
constexpr int kRevSlab = SCINT_REV_SLAB;     // most tau rows accumulated per workgroup: 1024 * 36 B = 36 KiB of LDS
constexpr int kRevCap = SCINT_REV_CAP;       // batched back-map: workgroups of a launch (each walks the launch's work items with this stride); 0 = one workgroup per item
constexpr int kRevThreadsK = 256;  // threads per workgroup
// Round 2 ran one 1024-thread workgroup with 144 KiB of LDS per Doppler column (all 4096 delay rows):
// alone 0.41 ms per 4096^2 image, but inside the chi^2 sweep 1.95 ms -- a workgroup that needs a whole
// CU never finds one while the mat-vec keeps two 64-KiB workgroups on every CU
// (profiles/r02_modeler_kernel_stats.csv).  A 256-thread / 36-KiB workgroup fits beside ONE mat-vec
// workgroup, i.e. it starts whenever any mat-vec workgroup retires.  A column then takes several
// workgroups (slabs of <= 1024 delay rows); each still walks all theta_i, but the delay a pair can reach
// is bounded from theta_i and the column's Doppler interval alone, so the lanes whose pairs cannot fall
// in the slab leave before they load anything (a slab is a contiguous run of i).

// ---- order-independent (bit-reproducible) accumulation ---------------------------------------
// Many (i, j) pairs fall in one CS pixel and arrive in scheduling order.  Plain float64 adds
// would make the sum depend on that order, so every addend x is first split on a fixed binary
// grid derived from an a-priori bound 2^E > |x|:
//     hi = x rounded to a multiple of 2^(E-36),   lo = (x - hi) rounded to a multiple of 2^(E-73)
// Up to 2^17 addends then sum EXACTLY in float64 in both accumulators (multiples of the grid,
// below 2^53 grid steps), so the two sums do not depend on the order, and hi_sum + lo_sum is one
// deterministic rounding.  What is dropped is < 2^(E-74) per addend: far below the float64
// rounding of any pixel that matters (E is within a few tens of binades of the largest pixel).
struct RevSplit {
    double s1, s2;     // 1.5 * 2^(E-36+52), 1.5 * 2^(E-73+52): (x + s) - s rounds x to the grid
    bool exact;        // false: degenerate bound (0, inf, NaN, extreme exponent) -> plain adds
};
__device__ inline RevSplit rev_split_for(double vmax, double min_dth, double two_eta, bool doubled) {
    RevSplit r;
    // |value / sqrt(|2 eta (th_i - th_j)|)| <= vmax / sqrt(|2 eta| min_dth); a Hermitian explicit
    // pixel adds w_ij + conj(w_ji): twice that
    const double bound = (doubled ? 2.0 : 1.0) * vmax / sqrt(fabs(two_eta) * min_dth);
    r.exact = isfinite(bound) && bound > 0.0;
    int e = 0;
    if (r.exact) { (void)frexp(bound, &e); r.exact = e > -900 && e < 900; }   // bound < 2^e
    r.s1 = ldexp(1.5, e + 16);
    r.s2 = ldexp(1.5, e - 21);
    return r;
}

// max |value| and the smallest theta spacing, as bit patterns of non-negative doubles (their
// integer order is their numeric order, so atomicMax / atomicMin are exact and deterministic).
// value: rank-1 -> |w| * max|v|^2 is formed by the consumer from max|v|; explicit -> max|thth_ij|.
__global__ void __launch_bounds__(256) rev_bound_kernel(RevParams p, unsigned long long* out) {
    __shared__ double red[4];
    const int64_t nvals = p.rank1 ? (int64_t)p.N : (int64_t)p.N * p.N;
    const cplx* __restrict__ src = p.rank1 ? p.vec : p.thth;
    double vmax = 0.0, dmin = INFINITY;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nvals; k += (int64_t)gridDim.x * 256) {
        const cplx v = gload(src + (p.rank1 ? k : (k / p.N) * p.ld + (k % p.N)));
        const double a = fmax(fabs(v.x), fabs(v.y));
        bad |= !(a == a);
        vmax = fmax(vmax, a);
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i + 1 < p.N; i += 256) {
            const double d = gload(p.th + i + 1) - gload(p.th + i);
            bad |= !(d > 0.0);
            dmin = fmin(dmin, d);
        }
    if (bad) { vmax = INFINITY; dmin = 0.0; }          // poisons the bound -> plain adds
    for (int o = 32; o > 0; o >>= 1) {
        vmax = fmax(vmax, __shfl_xor(vmax, o, 64));
        dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        vmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(out, (unsigned long long)__double_as_longlong(vmax * 1.4142135623730951));   // |v| <= sqrt2 max(|re|,|im|)
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (lane == 0) red[w] = dmin;
        __syncthreads();
        if (threadIdx.x == 0) {
            dmin = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
            atomicMin(out + 1, (unsigned long long)__double_as_longlong(dmin));
        }
    }
}

// first j in [0, N] with th[j] - thi >= lo (N if none): gallop out from the guess g, then bisect.
// fl(th[j] - thi) is non-decreasing in j, so the predicate is monotone.
__device__ inline int rev_first_ge(const double* th, int N, double thi, double lo, int g) {
    g = min(max(g, 0), N - 1);
    int a, b;  // answer in [a, b]; pred(a - 1) false or a == 0; pred(b) true or b == N
    if (gload(th + g) - thi >= lo) {
        b = g; a = 0;
        for (int step = 1; b - step >= 0; step <<= 1) {
            if (gload(th + (b - step)) - thi >= lo) { b -= step; }
            else { a = b - step + 1; break; }
        }
        if (a == 0 && b > 0 && !(gload(th) - thi >= lo)) a = 1;
    } else {
        a = g + 1; b = N;
        for (int step = 1; a - 1 + step < N; step <<= 1) {
            if (gload(th + (a - 1 + step)) - thi >= lo) { b = a - 1 + step; break; }
            a += step;
        }
    }
    while (a < b) {
        const int m = (a + b) >> 1;
        if (gload(th + m) - thi >= lo) b = m; else a = m + 1;
    }
    return a;
}

// Per-IMAGE constants, formed once by one thread instead of by each of the 65 536 wavefronts of a 4096^2 back-map
// (round 4: with the walk and the pair arithmetic switched off the kernel still took 0.116 of its 0.295 ms -- prologue,
// pre-pass and epilogue; the split grid alone is a frexp, two ldexp, a square root and a division, the window two more
// divisions): the split grid of rev_split_for, the mean theta spacing and the window width.  Same expressions, same bits.
constexpr int kRevWin = 8;   // widest candidate window (theta centres per fd bin, + 2)
// (the word indices kRevS1 .. kRevBandHi are in thth.hpp)
__device__ inline void rev_setup_consts(bool rank1, bool hermitian, const double* w, const double* th, int N, double two_eta,
                                        const GeomDev& g, double vmax, double dmin, unsigned long long* bound) {
    const double aw = rank1 ? fabs(gload(w)) : 0.0;
    const RevSplit sp = rev_split_for(rank1 ? aw * vmax * vmax : vmax, dmin, two_eta, !rank1 && hermitian);
    // window width from the mean theta spacing (the window START depends on the column and stays in the gather)
    const double th_step = N > 1 ? (gload(th + N - 1) - gload(th)) / (double)(N - 1) : 0.0;
    const int W = th_step > 0.0 ? (int)fmin(ceil(g.fd1_step / th_step) + 2.0, (double)kRevWin) : 0;
    bound[kRevS1] = (unsigned long long)__double_as_longlong(sp.s1);
    bound[kRevS2] = (unsigned long long)__double_as_longlong(sp.s2);
    bound[kRevExact] = sp.exact ? 1ull : 0ull;
    bound[kRevThStep] = (unsigned long long)__double_as_longlong(th_step);
    bound[kRevW] = (unsigned long long)(W < 0 ? 0 : W);
}
__device__ inline void rev_uniform_consts(double dev, const double* th, int N, double th_step, double eta, const GeomDev& g,
                                          unsigned long long* bound);
__device__ inline double rev_grid_deviation(const double* th, int N, double th_step, int first, int stride);
__global__ void __launch_bounds__(64) rev_setup_kernel(RevParams p, GeomDev g, unsigned long long* bound, int diag_enabled) {
    // which kernel forms the image: rev_diag_kernel on a uniform grid (rank-1 Hermitian), else rev_gather_kernel
    const bool diag = diag_enabled && p.rank1 && p.hermitian && p.N > 1;
    const double th_step = diag ? (gload(p.th + p.N - 1) - gload(p.th)) / (double)(p.N - 1) : 0.0;
    double dev = diag ? rev_grid_deviation(p.th, p.N, th_step, (int)threadIdx.x, 64) : INFINITY;
    for (int o = 32; o > 0; o >>= 1) dev = fmax(dev, __shfl_xor(dev, o, 64));
    if (threadIdx.x != 0) return;
    rev_setup_consts(p.rank1 != 0, p.hermitian != 0, p.w, p.th, p.N, p.two_eta, g, __longlong_as_double((long long)bound[0]),
                     __longlong_as_double((long long)bound[1]), bound);
    rev_uniform_consts(dev, p.th, p.N, th_step, p.eta, g, bound);
}

// One workgroup owns the `slab` tau rows [blockIdx.y*slab, ...) of ONE fd column of recov
// and gathers every theta-theta pixel that np.histogram2d would drop there
// (ththmod.py:207-262): for each i the j with fd_map[i, j] in the column form one short
// interval (th is increasing); their tau bin picks the row.  Weighted sums and counts
// accumulate in LDS, are divided and written once -- no global atomics, no zero-fill and no
// separate normalise pass over the [ntau, nfd] image.  (A workgroup's pixels are one COLUMN of the
// row-major image: 16-B stores 16 nfd bytes apart; the chi^2 sweep asks for the transposed image
// instead, which every workgroup writes as one contiguous run.)  The sums are order-independent (RevSplit
// above), so the image is bit-reproducible.
//
// Lanes run over i (coalesced).  On a (nearly) uniform theta grid the interval of lane i is
// i + s0 .. i + s0 + W - 1 for a column-wide s0: the lane fetches that window and one guard on
// either side with INDEPENDENT loads (no dependent search chain), checks that the guards bracket
// the interval -- fl(th[j] - th[i]) is non-decreasing in j, so a guard below the column and a
// guard beyond it prove nothing was missed -- and falls back to a galloping search otherwise.
//
// The Hermitian second pass of the reference (-fd, -tau, conj) puts the mirror of pixel
// (j, i) exactly where the direct image of (i, j) falls (negation is exact in floating
// point), so pixel (i, j) contributes  w_ij + conj(w_ji)  with count 2.  For the rank-1
// Hermitian model w_ji == conj(w_ij) exactly, and sum and count are both halved.
constexpr int kRevBlock = 256;      // lanes per block of the chunk pre-pass: a whole chunk (round 3 used 16-lane blocks so that every thread had
                                    // one; the hull argument below does not care about the block length, and with one block per chunk
                                    // three of the four wavefronts skip the pre-pass instead of issuing it)
constexpr int kRevLiveWords = 64;   // 32 chunks per word: N <= 524288 is pruned, beyond that every chunk is walked

template <int kRevThreads, bool RANK1, bool TABLE>
__device__ __forceinline__ void rev_gather_body(const RevParams& p, const GeomDev& g, const int64_t col_in, const int slab_index) {
    extern __shared__ __attribute__((aligned(16))) double rev_lds[];
    const int slab = p.slab;
    // rev_lds[0 .. 4 slab): real hi, real lo, imag hi, imag lo grids; then the counts.  Always
    // indexed off the __shared__ array itself so that every access stays an LDS instruction.
    uint32_t* cnt = (uint32_t*)(rev_lds + 4 * slab);
    // columns of 4 neighbouring workgroups of one XCD are adjacent, so their 16 B stores
    // complete 64 B lines in that XCD's L2
    int64_t col = col_in;
    if ((col | 31) < g.nfd) col = (col & ~(int64_t)31) + (col & 7) * 4 + ((col >> 3) & 3);
    const int64_t row0 = (int64_t)slab_index * slab;
    const int rows = (int)min((int64_t)slab, g.ntau - row0);
    for (int r = threadIdx.x; r < rows; r += kRevThreads) {
        rev_lds[r] = 0.0; rev_lds[slab + r] = 0.0; rev_lds[2 * slab + r] = 0.0; rev_lds[3 * slab + r] = 0.0;
        ((uint32_t*)(rev_lds + 4 * slab))[r] = 0u;
    }
    __syncthreads();

    const double lo = ((double)col - 0.5) * g.fd1_step + g.fd0;        // histogram edges of the column
    const double hi = ((double)(col + 1) - 0.5) * g.fd1_step + g.fd0;
    const bool last = (col == g.nfd - 1);                              // last bin is closed on the right
    const double aw = RANK1 ? fabs(gload(p.w)) : 0.0;
    // the image's constants (rev_setup_kernel): uniform loads
    RevSplit sp;
    sp.s1 = __longlong_as_double((long long)p.bound[kRevS1]);
    sp.s2 = __longlong_as_double((long long)p.bound[kRevS2]);
    sp.exact = p.bound[kRevExact] != 0ull;
    const int N = p.N;
    // window start relative to i (its width W comes with the constants), from the mean theta spacing
    const double th_step = __longlong_as_double((long long)p.bound[kRevThStep]);
    const double q0 = lo / th_step;
    const bool grid_ok = th_step > 0.0 && isfinite(q0) && fabs(q0) < 1e9;
    const int s0 = grid_ok ? (int)floor(q0) - 1 : 0;
    const int W = grid_ok ? (int)p.bound[kRevW] : 0;
    const bool usable = g.fd1_step > 0.0 && g.tau1_step > 0.0;
    const bool col_tab = TABLE && p.walk != nullptr && gload(p.walk_col + col) != 0;     // uniform over the workgroup

    const double inv_tstep = p.inv_tau1_step;
    const double row_lo = (double)row0 - 2.0, row_hi = (double)(row0 + rows) + 1.0;   // estimate of bin + 0.5
    auto beyond = [&](double x) { return last ? (x > hi) : (x >= hi); };
    // What pair (i, j) adds to the slab: ONE guarded region per test (slab estimate, exact bin) -- results that
    // leave a chain of early returns as a struct cost a move or a select per field and per exit
    // (profiles/r03_revmap_counters.txt: 58 % of the kernel's vector instructions were neither arithmetic nor index work).
    auto pair = [&](int i, int j, double th_i, double th_j) {
        const double y = p.eta * (th_j * th_j - th_i * th_i);          // tau_map[i, j] (ththmod.py:208-210)
        // cheap slab test first (two rows of slack cover the rounding of this estimate): the exact
        // bin, the weight and the loads are only paid for by the slab that owns the pixel
        const double est = (y - g.tau0) * inv_tstep;
        if (i == j || est < row_lo || est > row_hi) return;            // i == j lands in the poisoned centre bin
        const int bin = hist_bin_rcp(y, g.tau0, g.tau1_step, inv_tstep, (int)g.ntau);
        const int by = bin - (int)row0;
        if (bin < 0 || by < 0 || by >= rows) return;
        // thth / sqrt(|2 eta fd_map.T|): NumPy divides complex by real as v * (1/c)
        const double scl = rsqrt(fabs(p.two_eta * (th_i - th_j)));   // (1 / sqrt costs a division on top: 25 instructions against 10)
        double wr, wi;
        uint32_t c = 1u;
        if (RANK1) {
            const cplx o = mulc(gload(p.vec + i), gload(p.vec + j));   // outer(V, conj(V)) * |w|  (:312-313)
            wr = (o.x * aw) * scl; wi = (o.y * aw) * scl;
        } else {
            const cplx v = gload(p.thth + (int64_t)i * p.ld + j);
            wr = v.x * scl; wi = v.y * scl;
            if (p.hermitian) {
                const cplx u = gload(p.thth + (int64_t)j * p.ld + i);
                wr += u.x * scl; wi += -(u.y * scl);
                c = 2u;
            }
        }
        if (sp.exact) {
            const double rh = (wr + sp.s1) - sp.s1, ih = (wi + sp.s1) - sp.s1;
            atomicAdd(&rev_lds[by], rh);
            atomicAdd(&rev_lds[2 * slab + by], ih);
            atomicAdd(&rev_lds[slab + by], ((wr - rh) + sp.s2) - sp.s2);
            atomicAdd(&rev_lds[3 * slab + by], ((wi - ih) + sp.s2) - sp.s2);
        } else {
            atomicAdd(&rev_lds[by], wr);
            atomicAdd(&rev_lds[2 * slab + by], wi);
        }
        atomicAdd((uint32_t*)(rev_lds + 4 * slab) + by, c);
    };
    // (Summing the runs of lanes that hit one accumulator in registers first -- a segmented scan over the
    // wave, legal because the grid-split addends sum exactly in any association -- was measured in round 3:
    // 0.58 ms against 0.38 ms per 4096^2 image.  The kernel is bound by its fp64 arithmetic per pair
    // (exact bin, 1/sqrt, split), not by the LDS atomics.)
    const double lo_hi_min = fmin(lo, hi), lo_hi_max = fmax(lo, hi);
    // Every pair of theta_i in this column has x = th_j - th_i in [lo, hi] and y = eta x (2 th_i + x): a parabola
    // in x, extremal at the interval ends or at its vertex x = -th_i.  If that range of y misses the slab by more
    // than a row on either side of the slack `pair` already allows, no pair of theta_i lands here.  (NaNs compare
    // false: such a lane goes on.)
    auto delay_range = [&](double th_i, double& ymin, double& ymax) {
        const double ya = p.eta * (lo * (2.0 * th_i + lo)), yb = p.eta * (hi * (2.0 * th_i + hi));
        ymin = fmin(ya, yb); ymax = fmax(ya, yb);
        if (-th_i >= lo_hi_min && -th_i <= lo_hi_max) {
            const double yv = -(p.eta * (th_i * th_i));
            ymin = fmin(ymin, yv); ymax = fmax(ymax, yv);
        }
    };
    auto misses_slab = [&](double ymin, double ymax) {
        return (ymax - g.tau0) * inv_tstep < row_lo - 1.0 || (ymin - g.tau0) * inv_tstep > row_hi + 1.0;
    };
    // Which 256-lane chunks of theta_i can reach the slab at all?  A slab is reached by a contiguous run of i, and
    // four slabs of a column used to walk all N lanes each -- a quarter of the kernel's vector instructions
    // (profiles/r03_revmap_counters.txt).  One pass over BLOCKS of lanes instead (kRevBlock): for a fixed x the delay is
    // monotone in theta_i (also as computed: every operation of `delay_range` is monotone in its rounded operand),
    // so the range of any lane of a block lies in the hull of the ranges of the block's two end lanes -- plus the
    // vertex values a lane in between may add (theta^2 is monotone on either side of 0).  That needs theta
    // increasing, which the bound pre-pass has checked (its minimum spacing is 0 otherwise).
    __shared__ uint32_t live[kRevLiveWords];
    __shared__ double rcp_small[64];                                   // 1 / count of the epilogue (an IEEE division per PIXEL was a tenth of the kernel)
    if (threadIdx.x < 64) rcp_small[threadIdx.x] = 1.0 / (double)threadIdx.x;
    static_assert(kRevThreads % kRevBlock == 0, "a block of the pre-pass lies inside one chunk");
    const int nchunk = (N + kRevThreads - 1) / kRevThreads;
    const bool prune = usable && nchunk <= 32 * kRevLiveWords && __longlong_as_double((long long)p.bound[1]) > 0.0;
    if (threadIdx.x < kRevLiveWords) live[threadIdx.x] = prune ? 0u : ~0u;
    __syncthreads();
    if (prune) {
        for (int ia = (int)threadIdx.x * kRevBlock; ia < N; ia += kRevThreads * kRevBlock) {      // (one wavefront's work at N <= 16 384)
            const double ta = gload(p.th + ia), tb = gload(p.th + min(N - 1, ia + kRevBlock - 1));
            double ymin, ymax, ymin_b, ymax_b;
            delay_range(ta, ymin, ymax);
            delay_range(tb, ymin_b, ymax_b);
            ymin = fmin(ymin, ymin_b); ymax = fmax(ymax, ymax_b);
            if (-tb <= lo_hi_max && -ta >= lo_hi_min) {            // some theta of the block may have its vertex in the column
                const double va = -(p.eta * (ta * ta)), vb = -(p.eta * (tb * tb));
                ymin = fmin(ymin, fmin(va, vb)); ymax = fmax(ymax, fmax(va, vb));
                if (ta <= 0.0 && tb >= 0.0) { const double v0 = -(p.eta * 0.0); ymin = fmin(ymin, v0); ymax = fmax(ymax, v0); }
            }
            if (!misses_slab(ymin, ymax)) {
                const int c = ia / kRevThreads;
                atomicOr(&live[c >> 5], 1u << (c & 31));
            }
        }
    }
    __syncthreads();
    for (int base = 0; usable && base < N; base += kRevThreads) {      // trip count uniform over the workgroup
        const int chunk = base / kRevThreads;
        if (nchunk <= 32 * kRevLiveWords && !((live[chunk >> 5] >> (chunk & 31)) & 1u)) continue;
        const int i = base + (int)threadIdx.x;
        bool active = i < N;
        const double th_i = active ? gload(p.th + i) : 0.0;
        if (active) {
            double ymin, ymax;
            delay_range(th_i, ymin, ymax);
            if (misses_slab(ymin, ymax)) active = false;
        }
        if (__ballot(active) == 0ull) continue;                        // wave-uniform
        const int g0 = i + s0;                                         // first candidate
        if (TABLE && col_tab) {
            // the partners of theta_i in this column come from the crop's table (thth.hpp): bit k - 1 <-> j = g0 - 1 + k
            unsigned mask = active ? (unsigned)gload(p.walk + col * N + i) : 0u;
            while (__ballot(mask != 0u) != 0ull) {
                if (mask != 0u) {
                    const int k = __ffs((int)mask);
                    mask &= mask - 1u;
                    const int j = g0 - 1 + k;
                    pair(i, j, th_i, gload(p.th + j));
                }
            }
            continue;
        }
        // th[g0 - 1 .. g0 + W], every index clamped into the array and every lane loading (no predicate per position:
        // ten predicated loads were a fifth of the kernel's instructions).  A clamped GUARD is the array's end element:
        // if that is still outside the column it speaks for everything beyond it, and a guard position outside the
        // array passes by its index alone; a clamped CANDIDATE is discarded by its index below.
        double tj[kRevWin + 2];
#pragma unroll
        for (int k = 0; k < kRevWin + 2; ++k) {
            tj[k] = 0.0;
            if (k > W + 1) continue;                                   // uniform: nothing beyond the upper guard is loaded
            tj[k] = gload(p.th + min(max(g0 - 1 + k, 0), N - 1));
        }
        double t_hi = tj[1];                                           // tj[W + 1] without dynamic indexing
#pragma unroll
        for (int k = 2; k < kRevWin + 2; ++k) t_hi = (k == W + 1) ? tj[k] : t_hi;
        const bool below_ok = (g0 - 1 < 0) || !(tj[0] - th_i >= lo);
        const bool above_ok = (g0 + W >= N) || beyond(t_hi - th_i);
        const bool windowed = active && W > 0 && below_ok && above_ok;
        // This lane's pairs are the j of ONE run ja .. ja + nj - 1 (fl(th[j] - th_i) is non-decreasing in j): from the
        // window when its guards bracket the column, else from a search.  One loop then serves every lane -- its trip
        // count is the wave's longest run (two or three on the grids of the path) -- with ONE copy of `evaluate`
        // instead of one per window position; the run's theta values are re-read (they were just loaded).
        int ja = 0, nj = 0;
        if (windowed) {
            int kf = 0, kl = -1;
#pragma unroll
            for (int k = 1; k <= kRevWin; ++k) {
                if (k > W) break;                                      // W is uniform over the workgroup
                const int j = g0 - 1 + k;
                const double x = tj[k] - th_i;                         // fd_map[i, j]  (ththmod.py:207)
                if ((unsigned)j < (unsigned)N && x >= lo && !beyond(x)) {
                    if (kl < 0) kf = k;
                    kl = k;
                }
            }
            ja = g0 - 1 + kf;
            nj = kl < 0 ? 0 : kl - kf + 1;
        } else if (active) {                                           // irregular grid
            ja = rev_first_ge(p.th, N, th_i, lo, g0);
            int jb = ja;
            while (jb < N && !beyond(gload(p.th + jb) - th_i)) ++jb;
            nj = jb - ja;
        }
        for (int r = 0; __ballot(r < nj) != 0ull; ++r) {
            if (r < nj) {
                const int j = ja + r;
                const double th_j = gload(p.th + j);
                const double x = th_j - th_i;
                // (inside a window every position is tested on its own, as a position outside the column may sit
                // between two inside it when theta is not monotone to the last bit)
                if (!windowed || (x >= lo && !beyond(x))) pair(i, j, th_i, th_j);
            }
        }
    }
    __syncthreads();
    // recov = nan_to_num(sum / count); the bin that receives the i == j terms is NaN in the
    // reference (x/0 weights) and therefore 0 after nan_to_num.
    for (int r = threadIdx.x; r < rows; r += kRevThreads) {
        const int64_t o = (row0 + r) * g.nfd + col;
        cplx out = mk(0.0, 0.0);
        if (o != p.centre) {
            const uint32_t c = cnt[r];
            double scl;
            if (c < 64u) scl = rcp_small[c]; else scl = 1.0 / (double)c;
            out = mk(nan_to_num((rev_lds[r] + rev_lds[slab + r]) * scl),
                     nan_to_num((rev_lds[2 * slab + r] + rev_lds[3 * slab + r]) * scl));
        }
        gstore(p.recov + (p.transposed ? col * g.ntau + (row0 + r) : o), out);
    }
}

template <int kRevThreads, bool RANK1>
__global__ void __launch_bounds__(kRevThreads) rev_gather_kernel(RevParams p, GeomDev g) {
    if (RANK1 && p.hermitian && p.bound[kRevUniform] != 0ull) return;      // rev_diag_kernel's image
    rev_gather_body<kRevThreads, RANK1, false>(p, g, (int64_t)blockIdx.x, (int)blockIdx.y);
}

// ---- several curvatures per launch (chi^2 sweep; thth.hpp) -------------------------------------------------------------
// One workgroup per curvature: what rev_bound_kernel + rev_setup_kernel do for one image (the same maxima, the same expressions:
// the same bits), without atomics and memsets -- a rank-1 image has only N values to look at -- plus the delay band.
// Band: every pair has tau_map = eta (th_j^2 - th_i^2) with |tau_map| <= Y = |eta| max th^2 also as computed (a difference of
// two non-negative squares is at most the larger one, and every rounding involved is monotone); the histogram bin is monotone in
// its argument, so all pixels that receive a pair lie in the delay rows bin(-Y) .. bin(Y); one row of slack each side, then the
// hull with its own mirror image about tau = 0 (row ntau / 2 of the shifted axis), which is what the chi^2 kernel needs:
// the partner (-fd, -tau) of a pixel in the band is in the band.  A band that touches either end of the axis is the whole axis.
// Everything outside the band is exactly 0 in the full image (no pair: count 0 -> 0/0 -> nan_to_num -> 0).
__global__ void __launch_bounds__(256) rev_prep_batch_kernel(const RevJobDev* __restrict__ jobs, RevBatch b, GeomDev g) {
    __shared__ double red[3][4];
    const RevJobDev jb = jobs[b.job[blockIdx.x]];
    const int N = jb.N;
    double vmax = 0.0, dmin = INFINITY, t2max = 0.0;
    bool bad = false;
    for (int k = threadIdx.x; k < N; k += 256) {
        const cplx v = gload(jb.vec + k);
        const double a = fmax(fabs(v.x), fabs(v.y));
        bad |= !(a == a);
        vmax = fmax(vmax, a);
        const double t = gload(jb.th + k);
        t2max = fmax(t2max, t * t);
        bad |= !(t == t);
        if (k + 1 < N) {
            const double d = gload(jb.th + k + 1) - t;
            bad |= !(d > 0.0);
            dmin = fmin(dmin, d);
        }
    }
    if (bad) { vmax = INFINITY; dmin = 0.0; t2max = INFINITY; }   // poisons the bound -> plain adds; the whole axis
    for (int o = 32; o > 0; o >>= 1) {
        vmax = fmax(vmax, __shfl_xor(vmax, o, 64));
        dmin = fmin(dmin, __shfl_xor(dmin, o, 64));
        t2max = fmax(t2max, __shfl_xor(t2max, o, 64));
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = vmax; red[1][w] = dmin; red[2][w] = t2max; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    vmax = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3])) * 1.4142135623730951;   // |v| <= sqrt2 max(|re|, |im|)
    dmin = fmin(fmin(red[1][0], red[1][1]), fmin(red[1][2], red[1][3]));
    t2max = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    if (N < 2) dmin = __longlong_as_double(0x7f7f7f7f7f7f7f7fll);        // (what the memset of the one-image path leaves when no spacing exists)
    unsigned long long* bound = jb.bound;
    bound[0] = (unsigned long long)__double_as_longlong(vmax);
    bound[1] = (unsigned long long)__double_as_longlong(dmin);
    rev_setup_consts(true, true, jb.w, jb.th, N, jb.two_eta, g, vmax, dmin, bound);
    int64_t lo = 0, hi = g.ntau - 1;
    const double Y = fabs(jb.eta) * t2max;
    if (Y == Y && Y < INFINITY && g.tau1_step > 0.0) {
        const int64_t bl = hist_bin(-Y, g.tau0, g.tau1_step, g.ntau), bh = hist_bin(Y, g.tau0, g.tau1_step, g.ntau);
        // (-1: the value lies off the axis -- the band then runs to that end of it)
        lo = bl < 0 ? 0 : bl - 1;
        hi = bh < 0 ? g.ntau - 1 : bh + 1;
        const int64_t hq = g.ntau / 2;
        const int64_t mlo = 2 * hq - hi, mhi = 2 * hq - lo;
        lo = lo < mlo ? lo : mlo;
        hi = hi > mhi ? hi : mhi;
        if (lo <= 0 || hi >= g.ntau - 1) { lo = 0; hi = g.ntau - 1; }
    }
    bound[kRevBandLo] = (unsigned long long)lo;
    bound[kRevBandHi] = (unsigned long long)hi;
}

__global__ void __launch_bounds__(kRevThreadsK) rev_gather_batch_kernel(const RevJobDev* __restrict__ jobs, RevBatch b, GeomDev g, int slab,
                                                                         int nslab) {
    // work item w = (image, slab, Doppler column), columns fastest; a launch has one workgroup per item, or -- kRevCap -- fewer
    // that walk the items with the grid's stride
    const int64_t total = g.nfd * (int64_t)nslab * (int64_t)b.n;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t col = w % g.nfd;
        const int sl = (int)((w / g.nfd) % nslab), img = (int)(w / (g.nfd * (int64_t)nslab));
        const RevJobDev jb = jobs[b.job[img]];
        if (jb.bound[kRevUniform] != 0ull) continue;                       // rev_diag_batch_kernel's image
        // slabs that miss the curvature's delay band are skipped: nothing of them is read afterwards
        const int64_t row0 = (int64_t)sl * slab;
        if (row0 > (int64_t)jb.bound[kRevBandHi] || row0 + slab - 1 < (int64_t)jb.bound[kRevBandLo]) continue;
        RevParams p;
        p.thth = nullptr; p.ld = jb.N;
        p.vec = jb.vec; p.w = jb.w; p.rank1 = 1;
        p.th = jb.th; p.N = jb.N;
        p.eta = jb.eta; p.two_eta = jb.two_eta;
        p.hermitian = 1; p.slab = slab; p.centre = jb.centre;
        p.recov = b.recov[img]; p.transposed = 1;
        p.bound = jb.bound; p.inv_tau1_step = jb.inv_tau1_step;
        p.walk = jb.walk; p.walk_col = jb.walk_col;
        rev_gather_body<kRevThreadsK, true, true>(p, g, col, sl);
        __syncthreads();            // (the next item zeroes the accumulators this one has just read)
    }
}

// ---- the rank-1 Hermitian back-map on a UNIFORM theta grid: pairs along diagonals (round 6) ---------------------------------
// rev_gather_body lets lanes run over theta_i and FINDS the partners theta_j of every lane in its Doppler column -- with one
// partner per (theta_i, column) on the grids of the path that search (window, guards, run detection, or the partner table and its
// ballot loop) costs more than the pair's own arithmetic, and every delay slab repeats it: 93.5 M vector instructions per 4096^2
// image for ~13 M of pair arithmetic (profiles/r04_revmap_counters.txt).  On a uniform grid theta_k = theta_0 + k d (checked, not
// assumed: rev_uniform_consts below; every grid of the path is one -- centres of linspace edges, and a crop keeps a contiguous
// run of them) the pairs of a Doppler column are whole DIAGONALS j = i + s: fd_map = theta_j - theta_i is s d to within the
// measured deviation, so the column's candidates are the s with s d within that slack of [lo, hi) -- one to three of them, each a
// dense run of lanes with nothing to search -- and every pair still takes the column test on its own fl(theta_j - theta_i), the
// same comparison np.histogram2d makes.  Along a diagonal the delay eta (theta_j^2 - theta_i^2) = 2 eta s d^2 i + const is LINEAR
// in i, m = 2 |eta s| d^2 / dtau rows per step:
//   * a slab of delay rows is a contiguous run of i (closed form, two lanes of margin; the exact bin of every pair decides);
//   * m >= 1: consecutive pairs fall in DIFFERENT rows, so the lanes of a sweep never meet in an accumulator and the sum of a
//     pixel has a fixed order with plain float64 adds -- diagonal after diagonal (a barrier between them);  m >= 1/P (P <= 4):
//     P sweeps over i = p mod P, each with that property;
//   * flatter than that lanes still run densely over the pairs, every wavefront over one contiguous block of them: the bins
//     of 64 consecutive pairs are non-decreasing, so a row's pairs are a run of lanes, a segmented scan (a fixed tree, as
//     deep as the longest possible run) leaves the run's sum in its last lane, and that lane alone adds it to the row; the row a
//     block shares with the block before it is added last, block after block, behind a barrier.  (A first version gave every
//     delay row a thread that summed its run one pair after the other: the flattest diagonals -- thousands of pairs in six
//     rows -- made that a chain of dependent loads on six lanes, 0.36 .. 2.3 ms per image.)
// No order-independent split (eight adds and three of five LDS atomics per pair), no collisions (flat curvatures: +40 % of an
// image), no chunk pre-pass, 20 bytes of LDS per delay row instead of 36.  Counters of one 4096^2 image (mean of 0.25 / 1 / 4
// eta_true, profiles/r06_revmap_counters_first.txt, r06_revmap_slim_ab.txt): 89.0 M vector instructions in rev_gather_kernel,
// 41.2 M here (55.4 M before the per-diagonal geometry was cut to one pass without IEEE divisions and the strided sweeps lost
// their runtime `%`; r06_revmap_slim2_ab.txt); 246 -> 153 us.  What is left
// above the pairs' own ~13 M is lanes: a slab's share of a diagonal is 1024 / m lanes plus the margin, i.e. a little over one or
// two 256-lane sweeps with the last one mostly empty, and every wavefront repeats the uniform geometry.  The histogram bin is floor((y - tau0) / dtau + 1/2)
// whenever that argument is at least 1e-6 away from an integer (the edges of np.histogram2d round at 1e-12 of a row on any sane
// axis; rev_uniform_consts checks the axis), else hist_bin_rcp's exact edge comparisons.  The image agrees with the reference's
// to its rounding (each pixel is the same addends in another order; tests: 1e-9 of the peak against the oracle, bits pinned).
constexpr int kDiagStrideMax = 4;
#ifndef SCINT_DIAG_SLAB
#define SCINT_DIAG_SLAB 1024
#endif
constexpr int kDiagSlab = SCINT_DIAG_SLAB;    // delay rows per workgroup: 20 KiB of LDS at 1024

// Decides whether an image takes this kernel (rank-1, Hermitian) and leaves the slack of the column test.  `dev` is the largest
// |theta_k - (theta_0 + k step)| over the grid (the caller's reduction; NaN counts as infinite).
__device__ inline void rev_uniform_consts(double dev, const double* th, int N, double th_step, double eta, const GeomDev& g,
                                          unsigned long long* bound) {
    bool ok = N >= 2 && N < (1 << 22) && th_step > 0.0 && dev <= 1e-9 * th_step;
    double slack = 0.0;
    if (ok) {
        const double tmax = fmax(fabs(gload(th)), fabs(gload(th + N - 1)));
        slack = 2.0 * dev + 16.0 * 2.220446049250313e-16 * tmax;        // |fl(theta_j - theta_i) - s step| <= slack
        const double m1 = 2.0 * fabs(eta) * th_step * th_step / g.tau1_step;
        ok = tmax < INFINITY && g.tau1_step > 0.0 && g.fd1_step > 0.0 && m1 >= 1e-6 && m1 < INFINITY &&
             fabs(g.tau0) / g.tau1_step + (double)g.ntau < 1e8 && fabs(g.fd0) / g.fd1_step + (double)g.nfd < 1e8 &&
             tmax / th_step < 1e8;
    }
    bound[kRevUniform] = ok ? 1ull : 0ull;
    bound[kRevSlack] = (unsigned long long)__double_as_longlong(slack);
}
__device__ inline double rev_grid_deviation(const double* th, int N, double th_step, int first, int stride) {
    const double t0 = gload(th);
    double dev = 0.0;
    for (int k = first; k < N; k += stride) {
        const double e = fabs(gload(th + k) - (t0 + (double)k * th_step));
        dev = (e == e) ? fmax(dev, e) : INFINITY;
    }
    return dev;
}

template <int T, bool FUSE>
__device__ __forceinline__ void rev_diag_body(const RevParams& p, const GeomDev& g, const int64_t col_in, const int slab_index) {
    extern __shared__ __attribute__((aligned(16))) double rev_lds[];
    const int slab = p.slab;
    // rev_lds[0 .. slab): real sums, [slab .. 2 slab): imaginary sums, then the counts
    int64_t col = col_in;
    if ((col | 31) < g.nfd) col = (col & ~(int64_t)31) + (col & 7) * 4 + ((col >> 3) & 3);      // (as rev_gather_body)
    const int64_t row0 = (int64_t)slab_index * slab;
    const int rows = (int)min((int64_t)slab, g.ntau - row0);
    __shared__ double rcp_small[64];
    if (threadIdx.x < 64) rcp_small[threadIdx.x] = 1.0 / (double)threadIdx.x;

    const double lo = ((double)col - 0.5) * g.fd1_step + g.fd0;        // histogram edges of the column
    const double hi = ((double)(col + 1) - 0.5) * g.fd1_step + g.fd0;
    const bool last = (col == g.nfd - 1);                              // last bin is closed on the right
    const double aw = fabs(gload(p.w));
    const int N = p.N;
    const double d = __longlong_as_double((long long)p.bound[kRevThStep]);
    const double slack = __longlong_as_double((long long)p.bound[kRevSlack]);
    const double th0 = gload(p.th);
    const double inv_tstep = p.inv_tau1_step;
    const int ntau = (int)g.ntau;
    // FUSE relies on the histogram being mirror-symmetric -- pair (j, i) in pixel (nfd - c, ntau - r) when (i, j) is in (c, r):
    // -x and -y are exact, but the edges np.histogram2d computes are mirror images of each other only to ~1e-9 of a step
    // (step = fd[1] - fd[0] carries the rounding of fd[1]), and a value ON an edge belongs to one side.  Any pair within 1e-6 of
    // a step of an edge therefore has its mirrored pair binned as well; if that does not land in the mirrored pixel the
    // curvature's flag is raised and its chi^2 is formed again from a written image (fft.hip).  Column 0 and row 0 (whose
    // mirrors are off the axes) are not fused at all.
    const double col_tol = 1e-6 * g.fd1_step;
    bool on_edge = false;
    auto in_column = [&](double x) {
        const bool in = x >= lo && (last ? x <= hi : x < hi);
        if (FUSE && col != 0 && (fabs(x - lo) <= col_tol || fabs(x - hi) <= col_tol)) {
            const int64_t mc = g.nfd - col;
            const double mlo = ((double)mc - 0.5) * g.fd1_step + g.fd0, mhi = ((double)(mc + 1) - 0.5) * g.fd1_step + g.fd0;
            const bool mirrored = -x >= mlo && (mc == g.nfd - 1 ? -x <= mhi : -x < mhi);
            on_edge |= in != mirrored;
        }
        return in;
    };
    auto edge = [&](int k) { return ((double)k - 0.5) * g.tau1_step + g.tau0; };       // np.histogram2d's delay edges
    auto delay = [&](int i, int j) {
        const double a = gload(p.th + i), b = gload(p.th + j);
        return p.eta * (b * b - a * a);                                  // tau_map[i, j] (ththmod.py:208-210)
    };
    // weight of pair (i, j) if it lies in the column: thth / sqrt(|2 eta fd_map.T|), thth = outer(V, conj(V)) |w| (:312-313)
    auto weight = [&](int i, int j, double th_i, double th_j, double& wr, double& wi) {
        const double scl = rsqrt(fabs(p.two_eta * (th_i - th_j)));
        const cplx o = mulc(gload(p.vec + i), gload(p.vec + j));
        wr = (o.x * aw) * scl; wi = (o.y * aw) * scl;
    };
    // np.histogram2d's delay bin of y, -1 off the axis
    auto bin_of = [&](double y) {
        const double t = (y - g.tau0) * inv_tstep + 0.5;
        const double kf = floor(t), f = t - kf;
        if (f >= 1e-6 && f <= 1.0 - 1e-6) return (t >= 0.0 && t < (double)ntau) ? (int)kf : -1;
        const int k = hist_bin_rcp(y, g.tau0, g.tau1_step, inv_tstep, ntau);
        if (FUSE) {
            const int km = hist_bin_rcp(-y, g.tau0, g.tau1_step, inv_tstep, ntau);
            on_edge |= (k >= 1 && km != ntau - k) || (km >= 1 && k != ntau - km);
        }
        return k;
    };
    __shared__ double carry_re[T / 64], carry_im[T / 64];
    __shared__ uint32_t carry_c[T / 64];
    __shared__ int carry_by[T / 64];
    // candidate diagonals: s d within `slack` of the column (one more on either side when the quotient is within 1e-9 of an integer)
    const double qa = (lo - slack) / d, qb = (hi + slack) / d;
    int s_lo = (int)fmax(ceil(qa - 1e-9 * (1.0 + fabs(qa))), -(double)N);
    int s_hi = (int)fmin(floor(qb + 1e-9 * (1.0 + fabs(qb))), (double)N);
    s_lo = max(s_lo, -(N - 1)); s_hi = min(s_hi, N - 1);
    const double y_lo = edge((int)row0), y_hi = edge((int)row0 + rows);
    // Diagonal s against the slab (uniform over the workgroup, and every wavefront computes it for itself -- so it is kept short:
    // the first version formed it twice per diagonal with four IEEE divisions each, and 55 M vector instructions per image were
    // mostly this, profiles/r06_revmap_counters_first.txt): the pairs (i, i + s), i = i0 .. i0 + L - 1, have the delay
    // y(i) ~ A i + C; [ia, ib) are the lanes that can reach the slab, two of margin -- the exact bin of every pair decides, so the
    // reciprocal of A may be the hardware's approximation (1e-8 of 4096 lanes against a margin of two).
    struct Diag { int i0, L, ia, ib; double A, inv_m; };
    const double two_eta_d2 = 2.0 * p.eta * d * d;
    auto diag_of = [&](int s, Diag& q) {
        q.i0 = s < 0 ? -s : 0; q.L = N - (s < 0 ? -s : s);
        q.A = two_eta_d2 * (double)s;
        const double C = p.eta * ((double)s * d) * (2.0 * th0 + (double)s * d);
        const double inv_a = __builtin_amdgcn_rcp(q.A);
        q.inv_m = fabs(inv_a) * g.tau1_step;                           // steps of i per delay row
        const double fa = (y_lo - C) * inv_a, fb = (y_hi - C) * inv_a;
        const double f_min = fmin(fa, fb) - 2.0, f_max = fmax(fa, fb) + 2.0;
        q.ia = (int)fmin(fmax(floor(f_min), (double)q.i0), (double)(q.i0 + q.L));
        q.ib = (int)fmin(fmax(ceil(f_max) + 1.0, (double)q.i0), (double)(q.i0 + q.L));       // exclusive
        return s != 0 && q.ia < q.ib;                                  // (i == j lands in the poisoned centre bin)
    };
    for (int r = threadIdx.x; r < rows; r += T) {
        rev_lds[r] = 0.0; rev_lds[slab + r] = 0.0;
        ((uint32_t*)(rev_lds + 2 * slab))[r] = 0u;
    }
    __syncthreads();
    for (int s = s_lo; s <= s_hi; ++s) {                               // uniform over the workgroup
        Diag q;
        if (!diag_of(s, q)) continue;
        const int i0 = q.i0, L = q.L, ia = q.ia, ib = q.ib;
        const bool up = q.A > 0.0;
        const int P = (int)fmin(ceil(1.0102 * q.inv_m), 1e6);          // (a step of i is 1 / inv_m delay rows)
        if (P <= kDiagStrideMax) {
            // pairs of one sweep are P lanes apart: at least 1.01 delay rows -- no two of them in one accumulator
            // sweep `pass` takes the i = pass (mod P), whatever the slab's first lane: the order of a pixel's addends does not depend
            // on the slab.  (ia mod P without a division: P <= 4 -- a runtime `%` was a tenth of the kernel's instructions.)
            static_assert(kDiagStrideMax <= 4, "ia mod P below is written for P <= 4");
            const int ia_mod = P == 1 ? 0 : (P == 2 ? (ia & 1) : (P == 3 ? ia % 3 : (ia & 3)));
            for (int pass = 0; pass < P; ++pass) {
                const int first = pass - ia_mod + (pass < ia_mod ? P : 0);     // ia + first = pass (mod P), 0 <= first < P
                for (int i = ia + first + (int)threadIdx.x * P; i < ib; i += T * P) {
                    const int j = i + s;
                    const double th_i = gload(p.th + i), th_j = gload(p.th + j);
                    const double x = th_j - th_i;                      // fd_map[i, j]  (ththmod.py:207)
                    if (!in_column(x)) continue;
                    const int by = bin_of(p.eta * (th_j * th_j - th_i * th_i)) - (int)row0;
                    if ((unsigned)by >= (unsigned)rows) continue;
                    double wr, wi;
                    weight(i, j, th_i, th_j, wr, wi);
                    atomicAdd(&rev_lds[by], wr);
                    atomicAdd(&rev_lds[slab + by], wi);
                    atomicAdd((uint32_t*)(rev_lds + 2 * slab) + by, 1u);
                }
                __syncthreads();
            }
        } else {
            // flat diagonal: several pairs per delay row.  Lanes run densely over the pairs in the direction of increasing delay
            // (u), every wavefront over one contiguous block of them; the bins of a wavefront's 64 pairs are non-decreasing, so the
            // pairs of a row are a run of lanes: a segmented scan (fixed tree) leaves the run's sum in its last lane, which adds it
            // to the row -- one lane per row and instruction, and a wavefront's adds to a row it meets again in its next 64 pairs
            // are in program order.  The one row a block shares with the block before it (the bin of the pair in front of the
            // block) is kept aside and added after a barrier, block after block: every pixel's sum has ONE order.
            constexpr int NW = T / 64;
            const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
            const int ua = up ? ia - i0 : (i0 + L) - ib, ub = up ? ib - i0 : (i0 + L) - ia;
            const int B = (((ub - ua) + NW - 1) / NW + 63) & ~63;
            const int us = ua + wave * B, ue = min(ub, us + B);
            auto pair_i = [&](int u) { return up ? i0 + u : i0 + L - 1 - u; };
            const int run_max = (int)fmin(1.0102 * q.inv_m + 2.0, 64.0);   // pairs of one row among 64 consecutive ones: <= 1 / m + 1
            int k_first = INT_MIN;                                     // (no pair has this bin: the first block shares nothing)
            if (wave > 0 && us < ue) { const int i = pair_i(us - 1); k_first = bin_of(delay(i, i + s)); }
            if (lane == 0) { carry_re[wave] = 0.0; carry_im[wave] = 0.0; carry_c[wave] = 0u; carry_by[wave] = k_first == INT_MIN ? 0 : k_first - (int)row0; }
            for (int u0 = us; u0 < ue; u0 += 64) {                     // uniform over the wavefront
                const int u = u0 + lane;
                int k = INT_MAX;
                double wr = 0.0, wi = 0.0;
                uint32_t c = 0u;
                if (u < ue) {
                    const int i = pair_i(u), j = i + s;
                    const double th_i = gload(p.th + i), th_j = gload(p.th + j);
                    k = bin_of(p.eta * (th_j * th_j - th_i * th_i));
                    if (in_column(th_j - th_i) && (unsigned)(k - (int)row0) < (unsigned)rows) {
                        weight(i, j, th_i, th_j, wr, wi);
                        c = 1u;
                    }
                }
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    if (d < run_max) {                                 // (uniform) no run is longer: lane - d is in another row
                        const int kk = __shfl_up(k, d, 64);
                        const double ar = __shfl_up(wr, d, 64), ai = __shfl_up(wi, d, 64);
                        const uint32_t ac = __shfl_up(c, d, 64);
                        if (lane >= d && kk == k) { wr += ar; wi += ai; c += ac; }
                    }
                }
                const int kn = __shfl_down(k, 1, 64);
                if ((lane == 63 || kn != k) && c != 0u) {
                    const int by = k - (int)row0;
                    if (k == k_first) {
                        carry_re[wave] += wr; carry_im[wave] += wi; carry_c[wave] += c;
                    } else {
                        atomicAdd(&rev_lds[by], wr);
                        atomicAdd(&rev_lds[slab + by], wi);
                        atomicAdd((uint32_t*)(rev_lds + 2 * slab) + by, c);
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0)
                for (int w = 1; w < NW; ++w)
                    if (carry_c[w] != 0u) {
                        const int by = carry_by[w];
                        rev_lds[by] += carry_re[w]; rev_lds[slab + by] += carry_im[w];
                        ((uint32_t*)(rev_lds + 2 * slab))[by] += carry_c[w];
                    }
            __syncthreads();
        }
    }
    // recov = nan_to_num(sum / count), the poisoned centre 0 (as rev_gather_body)
    double chi = 0.0;
    for (int r = threadIdx.x; r < rows; r += T) {
        const int64_t o = (row0 + r) * g.nfd + col;
        cplx out = mk(0.0, 0.0);
        if (o != p.centre) {
            const uint32_t c = ((uint32_t*)(rev_lds + 2 * slab))[r];
            double scl;
            if (c < 64u) scl = rcp_small[c]; else scl = 1.0 / (double)c;
            out = mk(nan_to_num(rev_lds[r] * scl), nan_to_num(rev_lds[slab + r] * scl));
        }
        if (FUSE && col != 0 && row0 + r != 0) {
            // an interior pixel: fft2(model) there IS recov (its mirror pixel holds the conjugate), so its chi^2 term is local
            const int q = (int)row0 + r;
            if (q >= p.band_lo && q <= p.band_hi) {
                const cplx z = gload(p.spec + col * g.ntau + q);
                const double re = out.x - z.x, im = out.y - z.y;
                chi += re * re + im * im;
            }
        } else {
            gstore(p.recov + (p.transposed ? col * g.ntau + (row0 + r) : o), out);
        }
    }
    if (FUSE) {
        __shared__ double chi_red[T / 64];
        chi = block_sum(chi, chi_red);
        if (threadIdx.x == 0) gstore(p.partial, chi);
        if (on_edge) atomicOr(p.asym, 1);
    }
}

__global__ void __launch_bounds__(kRevThreadsK) rev_diag_kernel(RevParams p, GeomDev g) {
    if (p.bound[kRevUniform] == 0ull) return;                              // rev_gather_kernel's image
    rev_diag_body<kRevThreadsK, false>(p, g, (int64_t)blockIdx.x, (int)blockIdx.y);
}
template <bool FUSE>
__global__ void __launch_bounds__(kRevThreadsK) rev_diag_batch_kernel(const RevJobDev* __restrict__ jobs, RevBatch b, GeomDev g, int slab,
                                                                       int nslab, int stride_max, RevFuse fz) {
    const int64_t total = g.nfd * (int64_t)nslab * (int64_t)b.n;
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        const int64_t col = w % g.nfd;
        const int sl = (int)((w / g.nfd) % nslab), img = (int)(w / (g.nfd * (int64_t)nslab));
        const RevJobDev jb = jobs[b.job[img]];
        double* const partial = FUSE ? fz.partial + (int64_t)img * fz.partial_stride + (col * nslab + sl) : nullptr;
        const int64_t row0 = (int64_t)sl * slab;
        if (jb.bound[kRevUniform] == 0ull ||                               // (the host sorts the jobs; this only keeps the two kernels apart)
            row0 > (int64_t)jb.bound[kRevBandHi] || row0 + slab - 1 < (int64_t)jb.bound[kRevBandLo]) {
            if (FUSE && threadIdx.x == 0) gstore(partial, 0.0);
            continue;
        }
        RevParams p;
        p.spec = fz.spec; p.partial = partial; p.asym = FUSE ? fz.asym + b.job[img] : nullptr;
        p.band_lo = (int)jb.bound[kRevBandLo]; p.band_hi = (int)jb.bound[kRevBandHi];
        p.thth = nullptr; p.ld = jb.N;
        p.vec = jb.vec; p.w = jb.w; p.rank1 = 1;
        p.th = jb.th; p.N = jb.N;
        p.eta = jb.eta; p.two_eta = jb.two_eta;
        p.hermitian = 1; p.slab = slab; p.centre = jb.centre;
        p.recov = b.recov[img]; p.transposed = 1;
        p.bound = jb.bound; p.inv_tau1_step = jb.inv_tau1_step;
        p.walk = nullptr; p.walk_col = nullptr; p.stride_max = stride_max;
        rev_diag_body<kRevThreadsK, FUSE>(p, g, col, sl);
        __syncthreads();
    }
}
// the grid test of every curvature of a sweep, before it starts (the host sorts a tail batch by it)
__global__ void __launch_bounds__(256) rev_uniform_kernel(const RevJobDev* __restrict__ jobs, GeomDev g, int32_t* __restrict__ flags, int diag_enabled) {
    __shared__ double red[4];
    const RevJobDev jb = jobs[blockIdx.x];
    const int N = diag_enabled ? jb.N : 0;
    const double th_step = N > 1 ? (gload(jb.th + N - 1) - gload(jb.th)) / (double)(N - 1) : 0.0;      // rev_setup_consts
    double dev = N > 1 ? rev_grid_deviation(jb.th, N, th_step, (int)threadIdx.x, 256) : INFINITY;
    for (int o = 32; o > 0; o >>= 1) dev = fmax(dev, __shfl_xor(dev, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dev;
    __syncthreads();
    if (threadIdx.x != 0) return;
    dev = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    rev_uniform_consts(dev, jb.th, N, th_step, jb.eta, g, jb.bound);
    flags[blockIdx.x] = jb.bound[kRevUniform] ? 1 : 0;
}

RevParams make_rev_params(const cplx* thth, const cplx* vec, const double* w, int rank1, const double* th, int N,
                          const GeomDev& g, double eta, int hermitian, cplx* recov) {
    RevParams p;
    p.thth = thth; p.ld = N;
    p.vec = vec; p.w = w; p.rank1 = rank1;
    p.th = th; p.N = N;
    p.eta = eta; p.two_eta = 2 * eta;
    p.hermitian = hermitian;
    p.slab = 0;
    // the bin the i == j terms fall in (fd_map = 0, tau_map = eta*0)
    const int64_t cbx = hist_bin(0.0, g.fd0, g.fd1_step, g.nfd);
    const int64_t cby = hist_bin(eta * 0.0, g.tau0, g.tau1_step, g.ntau);
    p.centre = (cbx >= 0 && cby >= 0) ? cby * g.nfd + cbx : -1;
    p.recov = recov;
    p.transposed = 0;
    p.bound = nullptr;
    p.inv_tau1_step = 1.0 / g.tau1_step;
    p.walk = nullptr; p.walk_col = nullptr;
    p.stride_max = kDiagStrideMax;
    p.spec = nullptr; p.partial = nullptr; p.asym = nullptr; p.band_lo = 0; p.band_hi = 0;
    return p;
}

// SCINT_REV_DIAG=0 keeps every image on the general kernel (read per call: the A/B of profiles/r06_revmap_diag_ab.txt, and the
// tests that hold the general kernel's pinned bits on uniform grids)
// (slabs of 512 / 1024 / 2048 rows and strided sweeps up to 1 / 2 / 4 / 8 passes were run as environment knobs, since removed:
// profiles/r06_revmap_diag_knobs.txt -- 512 rows 902 eta/s, 2048 rows 922, 1024 rows 937-940 whatever the stride limit)
static int diag_slab() { return kDiagSlab; }
static int diag_stride_max() { return kDiagStrideMax; }
static bool rev_diag_enabled() {
    const char* e = getenv("SCINT_REV_DIAG");
    return !(e && e[0] == '0');
}

// Enqueue the back-map: bound pre-pass (max |value|, min theta spacing) + the column gather.
int32_t launch_rev_map(RevParams p, const GeomDev& g, unsigned long long* bound /*[8] device scratch*/,
                       hipStream_t stream) {
    // bound[0] = 0.0 (max), bound[1] = 0x7f7f7f7f7f7f7f7f = 1.4e306 (min: above any spacing) as bit patterns,
    // written by the device (no host buffer involved: a pageable source would make this an in-line staged
    // copy on the tail stream)
    SCINT_HIP(hipMemsetAsync(bound, 0, 8, stream));
    SCINT_HIP(hipMemsetAsync((char*)bound + 8, 0x7f, 8, stream));
    const int64_t nvals = p.rank1 ? (int64_t)p.N : (int64_t)p.N * p.N;
    const unsigned nblk = (unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(nvals, 256 * 8)));
    hipLaunchKernelGGL(rev_bound_kernel, dim3(nblk), dim3(256), 0, stream, p, bound);
    hipLaunchKernelGGL(rev_setup_kernel, dim3(1), dim3(64), 0, stream, p, g, bound, rev_diag_enabled() ? 1 : 0);
    p.bound = bound;
    // equal slabs of at most kRevSlab delay rows
    p.slab = (int)ceil_div(g.ntau, ceil_div(g.ntau, (int64_t)kRevSlab));
    dim3 grid((unsigned)g.nfd, (unsigned)ceil_div(g.ntau, p.slab));
    SCINT_REQUIRE(grid.y <= 65535, "rev_map: ntau too large");
    if (p.rank1 && p.hermitian) {             // the uniform-grid kernel or the general one: the device decides (rev_setup_kernel), the other leaves at once
        RevParams q = p;
        q.slab = (int)ceil_div(g.ntau, ceil_div(g.ntau, (int64_t)diag_slab()));
        q.stride_max = diag_stride_max();
        dim3 qgrid((unsigned)g.nfd, (unsigned)ceil_div(g.ntau, q.slab));
        SCINT_REQUIRE(qgrid.y <= 65535, "rev_map: ntau too large");
        hipLaunchKernelGGL(rev_diag_kernel, qgrid, dim3(kRevThreadsK), (size_t)q.slab * 20, stream, q, g);
    }
    if (p.rank1)
        hipLaunchKernelGGL((rev_gather_kernel<kRevThreadsK, true>), grid, dim3(kRevThreadsK), (size_t)p.slab * 36, stream, p, g);
    else
        hipLaunchKernelGGL((rev_gather_kernel<kRevThreadsK, false>), grid, dim3(kRevThreadsK), (size_t)p.slab * 36, stream, p, g);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

int32_t launch_rev_map_rank1(const cplx* vec, const double* w, const double* th, int64_t N, const GeomDev& g,
                             double eta, cplx* recov, bool transposed, void* scratch, hipStream_t stream) {
    RevParams p = make_rev_params(nullptr, vec, w, 1, th, (int)N, g, eta, 1, recov);
    p.transposed = transposed ? 1 : 0;
    return launch_rev_map(p, g, (unsigned long long*)scratch, stream);
}

RevJobDev make_rev_job(const cplx* vec, const double* w, const double* th, int64_t N, const GeomDev& g, double eta,
                       unsigned long long* bound) {
    const RevParams p = make_rev_params(nullptr, vec, w, 1, th, (int)N, g, eta, 1, nullptr);
    RevJobDev j;
    j.vec = vec; j.w = w; j.th = th;
    j.eta = eta; j.two_eta = p.two_eta; j.inv_tau1_step = p.inv_tau1_step;
    j.centre = p.centre; j.bound = bound; j.N = (int32_t)N; j.pad = 0;
    j.walk = nullptr; j.walk_col = nullptr;
    return j;
}

// The window walk of rev_gather_body for every (column, theta_i) of one crop, without the curvature (thth.hpp).  Same
// expressions as the kernel (window start s0 and width W from the mean theta spacing, clamped unconditional loads, guards).
__global__ void __launch_bounds__(256) rev_walk_table_kernel(const double* __restrict__ th, int N, GeomDev g, uint8_t* __restrict__ masks,
                                                             uint8_t* __restrict__ col_ok) {
    __shared__ int bad;
    const int64_t col = blockIdx.x;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const double lo = ((double)col - 0.5) * g.fd1_step + g.fd0;
    const double hi = ((double)(col + 1) - 0.5) * g.fd1_step + g.fd0;
    const bool last = (col == g.nfd - 1);
    const double th_step = N > 1 ? (gload(th + N - 1) - gload(th)) / (double)(N - 1) : 0.0;        // rev_setup_consts
    const int W0 = th_step > 0.0 ? (int)fmin(ceil(g.fd1_step / th_step) + 2.0, (double)kRevWin) : 0;
    const double q0 = lo / th_step;
    const bool grid_ok = th_step > 0.0 && isfinite(q0) && fabs(q0) < 1e9;
    const int s0 = grid_ok ? (int)floor(q0) - 1 : 0;
    const int W = grid_ok ? (W0 < 0 ? 0 : W0) : 0;
    auto beyond = [&](double x) { return last ? (x > hi) : (x >= hi); };
    bool ok = true;
    for (int i = (int)threadIdx.x; i < N; i += 256) {
        const double th_i = gload(th + i);
        const int g0 = i + s0;
        double tj[kRevWin + 2];
#pragma unroll
        for (int k = 0; k < kRevWin + 2; ++k) {
            tj[k] = 0.0;
            if (k > W + 1) continue;
            tj[k] = gload(th + min(max(g0 - 1 + k, 0), N - 1));
        }
        double t_hi = tj[1];
#pragma unroll
        for (int k = 2; k < kRevWin + 2; ++k) t_hi = (k == W + 1) ? tj[k] : t_hi;
        const bool below_ok = (g0 - 1 < 0) || !(tj[0] - th_i >= lo);
        const bool above_ok = (g0 + W >= N) || beyond(t_hi - th_i);
        const bool windowed = W > 0 && below_ok && above_ok;
        unsigned mask = 0u;
        if (windowed) {
#pragma unroll
            for (int k = 1; k <= kRevWin; ++k) {
                if (k > W) break;
                const int j = g0 - 1 + k;
                const double x = tj[k] - th_i;
                if ((unsigned)j < (unsigned)N && x >= lo && !beyond(x)) mask |= 1u << (k - 1);
            }
        } else {
            ok = false;
        }
        masks[col * N + i] = (uint8_t)mask;
    }
    if (!ok) atomicOr(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) col_ok[col] = bad ? 0 : 1;
}

int32_t launch_rev_walk_table(const double* th, int64_t N, const GeomDev& g, uint8_t* masks, uint8_t* col_ok, hipStream_t stream) {
    SCINT_REQUIRE(th && masks && col_ok && N >= 1 && N < ((int64_t)1 << 31), "rev_map walk table: bad arguments");
    hipLaunchKernelGGL(rev_walk_table_kernel, dim3((unsigned)g.nfd), dim3(256), 0, stream, th, (int)N, g, masks, col_ok);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

int32_t launch_rev_map_rank1_batch(const RevJobDev* jobs_dev, const RevBatch& b, const GeomDev& g, const uint8_t* uniform, const RevFuse* fuse,
                                   RevBatch* general_out, RevBatch* uniform_out, hipStream_t stream) {
    SCINT_REQUIRE(b.n >= 1 && b.n <= kRevBatchMax, "rev_map batch: bad count");
    hipLaunchKernelGGL(rev_prep_batch_kernel, dim3((unsigned)b.n), dim3(256), 0, stream, jobs_dev, b, g);
    // the batch in two: the curvatures on a uniform grid (rev_uniform_kernel's flags, read back before the sweep) and the others
    RevBatch part[2];
    for (int k = 0; k < 2; ++k) {
        part[k].n = 0; part[k].pad = 0;
        for (int q = 0; q < kRevBatchMax; ++q) { part[k].job[q] = 0; part[k].recov[q] = nullptr; }
    }
    for (int q = 0; q < b.n; ++q) {
        RevBatch& dst = part[(uniform && uniform[b.job[q]]) ? 1 : 0];
        dst.job[dst.n] = b.job[q]; dst.recov[dst.n] = b.recov[q];
        ++dst.n;
    }
    for (int k = 0; k < 2; ++k) {
        if (part[k].n == 0) continue;
        const int slab = (int)ceil_div(g.ntau, ceil_div(g.ntau, (int64_t)(k ? diag_slab() : kRevSlab)));     // the slabs of launch_rev_map
        const int nslab = (int)ceil_div(g.ntau, slab);
        const int64_t total = g.nfd * (int64_t)nslab * (int64_t)part[k].n;
        SCINT_REQUIRE(total < ((int64_t)1 << 31), "rev_map batch: too many work items");
        const unsigned grid = (unsigned)(kRevCap > 0 ? std::min<int64_t>(total, kRevCap) : total);
        if (k && fuse) {
            SCINT_REQUIRE(fuse->spec && fuse->partial && fuse->asym && fuse->partial_stride >= g.nfd * (int64_t)nslab, "rev_map batch: bad fuse arguments");
            hipLaunchKernelGGL(rev_diag_batch_kernel<true>, dim3(grid), dim3(kRevThreadsK), (size_t)slab * 20, stream, jobs_dev, part[k], g, slab, nslab,
                               diag_stride_max(), *fuse);
        } else if (k) {
            hipLaunchKernelGGL(rev_diag_batch_kernel<false>, dim3(grid), dim3(kRevThreadsK), (size_t)slab * 20, stream, jobs_dev, part[k], g, slab, nslab,
                               diag_stride_max(), RevFuse{nullptr, nullptr, 0, nullptr});
        } else {
            hipLaunchKernelGGL(rev_gather_batch_kernel, dim3(grid), dim3(kRevThreadsK), (size_t)slab * 36, stream, jobs_dev, part[k], g, slab, nslab);
        }
    }
    SCINT_LAUNCH_CHECK();
    if (general_out) *general_out = part[0];
    if (uniform_out) *uniform_out = part[1];
    return SCINT_OK;
}
int64_t rev_diag_items_for(int64_t ntau, int64_t nfd) {
    const int slab = (int)ceil_div(ntau, ceil_div(ntau, (int64_t)diag_slab()));
    return nfd * ceil_div(ntau, slab);
}
int64_t rev_diag_items(const GeomDev& g) { return rev_diag_items_for(g.ntau, g.nfd); }
int32_t launch_rev_uniform(const RevJobDev* jobs_dev, int64_t njobs, const GeomDev& g, int32_t* flags_dev, hipStream_t stream) {
    SCINT_REQUIRE(jobs_dev && flags_dev && njobs >= 1 && njobs < ((int64_t)1 << 31), "rev_map grid test: bad arguments");
    hipLaunchKernelGGL(rev_uniform_kernel, dim3((unsigned)njobs), dim3(256), 0, stream, jobs_dev, g, flags_dev, rev_diag_enabled() ? 1 : 0);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

}  // namespace scint

using namespace scint;

extern "C" int32_t scint_thth_map(const scint_c128* cs, const scint_cs_geom* geom,
                                  const double* th_cents, int64_t M, const int32_t* keep_idx,
                                  int64_t N, double eta, int32_t hermitian, scint_c128* thth_out,
                                  void* stream_) {
    SCINT_REQUIRE(cs && geom && th_cents && keep_idx && thth_out, "thth_map: null pointer");
    SCINT_REQUIRE(M >= 1 && N >= 0 && N <= M, "thth_map: bad sizes");
    SCINT_REQUIRE(geom->dtau > 0 && geom->dfd > 0, "thth_map: tau and fd must be increasing");
    if (N == 0) return SCINT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    GatherJob job;
    job.eta = eta;
    job.two_eta = 2 * eta;
    job.keep = keep_idx;
    job.n = (int32_t)N;
    job.hermitian = hermitian;
    job.out = (cplx*)thth_out;
    job.ld = N;
    return launch_gather((const cplx*)cs, to_dev(*geom), th_cents, M, job, stream);
}

extern "C" int32_t scint_rev_map_workspace_bytes(size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "rev_map_workspace_bytes: null output");
    *bytes = 256;
    return SCINT_OK;
}

extern "C" int32_t scint_rev_map(const scint_c128* thth, const scint_c128* vec, const double* w,
                                 int32_t rank1, const double* th_cents, int64_t N,
                                 const scint_cs_geom* geom, double eta, int32_t hermitian,
                                 scint_c128* recov_out, void* workspace, size_t workspace_bytes,
                                 void* stream_) {
    SCINT_REQUIRE(geom && th_cents && recov_out && workspace, "rev_map: null pointer");
    SCINT_REQUIRE(workspace_bytes >= 256, "rev_map: workspace too small");
    SCINT_REQUIRE(rank1 ? (vec && w) : (thth != nullptr), "rev_map: missing input");
    SCINT_REQUIRE(N >= 1, "rev_map: bad N");
    hipStream_t stream = (hipStream_t)stream_;
    const GeomDev g = to_dev(*geom);
    return launch_rev_map(make_rev_params((const cplx*)thth, (const cplx*)vec, w, rank1, th_cents, (int)N, g, eta,
                                          hermitian, (cplx*)recov_out),
                          g, (unsigned long long*)workspace, stream);
}


// ------------------------------------------------------------------------------
// scint_sweep_keep: the crop of thth_redmap (ththmod.py:153-155) for a whole sweep, on the device
// ------------------------------------------------------------------------------
// keep[e] = ascending indices i with  th[i]^2 * eta_e < tau_max  and  |th[i]| < fd_half  -- NumPy's expression
// with its roundings (two multiplications, no contraction: this unit is built with -ffp-contract=off).
// One 256-thread workgroup per curvature; left-packed by a fixed-order block scan.  On the host this is a
// Python loop over the curvatures plus a 4 MB upload per sweep (3 % of a 4096^2 / 256-eta step).
struct KeepEtas { double eta[256]; };
__global__ void __launch_bounds__(256) sweep_keep_kernel(const double* __restrict__ th, int M, KeepEtas etas, int e0,
                                                         int neta, double tau_max, double fd_half,
                                                         int32_t* __restrict__ keep_idx, int32_t* __restrict__ keep_n) {
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    const int e = e0 + (int)blockIdx.x;
    if (e >= neta) return;
    const double eta = etas.eta[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int32_t* __restrict__ out = keep_idx + (int64_t)e * M;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        bool k = false;
        if (i < M) {
            const double t = th[i];
            const double t2 = t * t;
            k = (t2 * eta < tau_max) && (fabs(t) < fd_half);
        }
        const unsigned long long m = __ballot(k);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[w] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int ww = 0; ww < w; ++ww) off += wave_tot[ww];
        if (k) out[off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) keep_n[e] = base_s;
}

extern "C" int32_t scint_sweep_keep(const double* th_cents, int64_t M, const double* etas, int64_t neta,
                                    double tau_max, double fd_half, int32_t* keep_idx, int32_t* keep_n,
                                    void* stream_) {
    SCINT_REQUIRE(th_cents && etas && keep_idx && keep_n, "sweep_keep: null pointer");
    SCINT_REQUIRE(M >= 1 && M < (1 << 30) && neta >= 1, "sweep_keep: bad shape");
    hipStream_t stream = (hipStream_t)stream_;
    for (int64_t e0 = 0; e0 < neta; e0 += 256) {
        KeepEtas ke;
        const int nb = (int)std::min<int64_t>(256, neta - e0);
        for (int i = 0; i < 256; ++i) ke.eta[i] = i < nb ? etas[e0 + i] : 0.0;
        hipLaunchKernelGGL(sweep_keep_kernel, dim3((unsigned)nb), dim3(256), 0, stream, th_cents, (int)M, ke, (int)e0,
                           (int)neta, tau_max, fd_half, keep_idx, keep_n);
    }
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}
