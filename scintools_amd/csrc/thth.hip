// thth.hip -- CS -> theta-theta gather (thth_map + thth_redmap, ththmod.py:56-173)
// and theta-theta -> CS scatter (rev_map, ththmod.py:176-271).
//
// This translation unit is compiled with -ffp-contract=off: the bin an element
// falls in is decided by a floor / an ordered comparison of float64 expressions,
// and the reference evaluates those expressions with one IEEE rounding per NumPy
// ufunc.  A fused multiply-add would flip pixels.
#include <float.h>
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
// One 32 x 32 quadrant of a 64 x 64 packed tile per 256-thread block (32 lanes x 8 rows, four
// rows per thread): the quadrant's footprint in the CS is a compact patch (~16 f_D bins wide),
// which keeps the scattered 16-B reads inside few cache lines per wave.
__global__ void __launch_bounds__(256)
thth_gather_packed_kernel(const GeomDev* __restrict__ geoms, int64_t M,
                          const PackedJob* __restrict__ jobs, const int32_t* __restrict__ slots) {
    const PackedJob* __restrict__ jp = jobs + slots[blockIdx.z];
    const cplx* __restrict__ cs = jp->cs;
    const double* __restrict__ th = jp->th;
    const GeomDev g = geoms[jp->geom];
    const int I = blockIdx.y >> 1, J = blockIdx.x >> 1;       // 64-tile coordinates
    const int qi = blockIdx.y & 1, qj = blockIdx.x & 1;       // quadrant inside the tile
    const int nb = jp->nb, n = jp->n;
    if (I >= nb || J >= nb || J < I) return;
    const int32_t* __restrict__ keep = jp->keep;
    const double eta = jp->eta, two_eta = jp->two_eta;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int cj = qj * 32 + tx;                              // column inside the 64-tile
    const int j = J * kTB + cj;
    const int kj = j < n ? gload(keep + j) : 0;
    const double th_j = j < n ? gload(th + kj) : 0.0;
    cplx* __restrict__ tile = jp->tiles + (tile_offset(nb, I) + (J - I)) * kTileElems;
    // Three phases so that a lane's scattered CS reads are all in flight together:
    // (1) index math, (2) the loads, (3) weights and stores.
    int64_t off[4];
    double wgt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ri = qi * 32 + ty + 8 * r;
        const int i = I * kTB + ri;
        off[r] = -1;
        wgt[r] = 0.0;
        if (i < n && j < n && i != j) {
            const int ki = gload(keep + i);
            const double th_i = gload(th + ki);
            // upper element (i < j) reads (theta2 = th_i, theta1 = th_j); the lower half of a
            // diagonal tile is the conjugate of the mirrored upper element
            const double t2 = i < j ? th_i : th_j, t1 = i < j ? th_j : th_i;
            int64_t o = thth_offset(g, eta, t2, t1);
            if ((int64_t)ki + kj == M - 1) o = -1;          // anti-diagonal (ththmod.py:113)
            off[r] = o;
            wgt[r] = sqrt(fabs(two_eta * (t2 - t1)));
            if (i > j) wgt[r] = -wgt[r];                    // sign carries "conjugate"
        }
    }
    cplx val[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) val[r] = gload(cs + (off[r] >= 0 ? off[r] : 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cplx v = mk(0.0, 0.0);
        if (off[r] >= 0) {
            const double aw = fabs(wgt[r]);
            v = mk(val[r].x * aw, val[r].y * aw);
            if (wgt[r] < 0.0) v = conj(v);
            v = mk(nan_to_num(v.x), nan_to_num(v.y));
        }
        gstore_nt(tile + (qi * 32 + ty + 8 * r) * kTB + cj, v);   // written once, read much later
    }
}

int32_t launch_gather_packed(const GeomDev* geoms_dev, int64_t M, const PackedJob* jobs_dev,
                             const int32_t* slots_dev, int njobs, int nbmax, hipStream_t stream) {
    if (njobs <= 0 || nbmax <= 0) return SCINT_OK;
    SCINT_REQUIRE(nbmax <= 32767 && njobs <= 65535, "gather: grid too large");
    const int slot = profiler().begin(kProfGather, stream);
    hipLaunchKernelGGL(thth_gather_packed_kernel, dim3(2u * (unsigned)nbmax, 2u * (unsigned)nbmax, (unsigned)njobs),
                       dim3(256), 0, stream, geoms_dev, M, jobs_dev, slots_dev);
    profiler().end(kProfGather, slot, stream);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// rev_map
// ------------------------------------------------------------------------------
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

struct RevParams {
    const cplx* thth; int64_t ld;      // explicit matrix (rank1 == 0)
    const cplx* vec; const double* w;  // rank-1: |w| v v^H (rank1 == 1)
    int rank1;
    const double* th; int N;
    double eta, two_eta;
    int hermitian;
    int slab;         // tau rows per workgroup
    int64_t centre;   // flat index of the pixel the i == j terms poison, or -1
    cplx* recov;      // [ntau, nfd]
};

constexpr int kRevSlab = 2048;  // tau rows accumulated per workgroup: 2048 * 20 B = 40 KB of LDS

// smallest j in [0, N] with th[j] - thi >= lo (N if none), galloping out from the guess g.
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

// One workgroup owns the `slab` tau rows [blockIdx.y*slab, ...) of ONE fd column of recov
// and gathers every theta-theta pixel that np.histogram2d would drop there
// (ththmod.py:207-262): for each i the j with fd_map[i, j] in the column form one short
// interval (th is increasing); their tau bin picks the row.  Weighted sums and counts
// accumulate in LDS (ds_add_f64), are divided and written once -- no global atomics, no
// zero-fill and no separate normalise pass over the [ntau, nfd] image.
//
// The Hermitian second pass of the reference (-fd, -tau, conj) puts the mirror of pixel
// (j, i) exactly where the direct image of (i, j) falls (negation is exact in floating
// point), so pixel (i, j) contributes  w_ij + conj(w_ji)  with count 2.  For the rank-1
// Hermitian model w_ji == conj(w_ij) exactly, and sum and count are both halved.
__global__ void __launch_bounds__(256) rev_gather_kernel(RevParams p, GeomDev g) {
    extern __shared__ double rev_lds[];
    const int slab = p.slab;
    double* acc_re = rev_lds;
    double* acc_im = rev_lds + slab;
    uint32_t* cnt = (uint32_t*)(rev_lds + 2 * slab);
    // columns of 4 neighbouring workgroups of one XCD are adjacent, so their 16 B stores
    // complete 64 B lines in that XCD's L2
    int64_t col = blockIdx.x;
    if ((col | 31) < g.nfd) col = (col & ~(int64_t)31) + (col & 7) * 4 + ((col >> 3) & 3);
    const int64_t row0 = (int64_t)blockIdx.y * slab;
    const int rows = (int)min((int64_t)slab, g.ntau - row0);
    for (int r = threadIdx.x; r < rows; r += 256) { acc_re[r] = 0.0; acc_im[r] = 0.0; cnt[r] = 0u; }
    __syncthreads();

    const double lo = ((double)col - 0.5) * g.fd1_step + g.fd0;        // histogram edges of the column
    const double hi = ((double)(col + 1) - 0.5) * g.fd1_step + g.fd0;
    const bool last = (col == g.nfd - 1);                              // last bin is closed on the right
    const double aw = p.rank1 ? fabs(gload(p.w)) : 0.0;
    // start of the neighbour search: offset of the column centre in mean theta spacings
    const double th_step = p.N > 1 ? (gload(p.th + p.N - 1) - gload(p.th)) / (double)(p.N - 1) : 0.0;
    const double est = th_step > 0.0 ? 0.5 * (lo + hi) / th_step : 0.0;
    const int shift = (int)fmin(fmax(rint(est), -(double)p.N), (double)p.N);
    const bool usable = g.fd1_step > 0.0 && g.tau1_step > 0.0;
    for (int i = threadIdx.x; usable && i < p.N; i += 256) {
        const double th_i = gload(p.th + i);
        int j = rev_first_ge(p.th, p.N, th_i, lo, i + shift);
        for (; j < p.N; ++j) {
            const double th_j = gload(p.th + j);
            const double x = th_j - th_i;                              // fd_map[i, j]  (ththmod.py:207)
            if (last ? (x > hi) : (x >= hi)) break;
            if (i == j) continue;                                      // lands in the poisoned centre bin
            const double y = p.eta * (th_j * th_j - th_i * th_i);      // tau_map[i, j] (ththmod.py:208-210)
            const int64_t by = hist_bin(y, g.tau0, g.tau1_step, g.ntau) - row0;
            if (by < 0 || by >= rows) continue;
            // thth / sqrt(|2 eta fd_map.T|): NumPy divides complex by real as v * (1/c)
            const double scl = 1.0 / sqrt(fabs(p.two_eta * (th_i - th_j)));
            double wr, wi;
            uint32_t c = 1u;
            if (p.rank1) {
                const cplx o = mulc(gload(p.vec + i), gload(p.vec + j));  // outer(V, conj(V)) * |w|  (:312-313)
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
            atomicAdd(&acc_re[by], wr);
            atomicAdd(&acc_im[by], wi);
            atomicAdd(&cnt[by], c);
        }
    }
    __syncthreads();
    // recov = nan_to_num(sum / count); the bin that receives the i == j terms is NaN in the
    // reference (x/0 weights) and therefore 0 after nan_to_num.
    for (int r = threadIdx.x; r < rows; r += 256) {
        const int64_t o = (row0 + r) * g.nfd + col;
        cplx out = mk(0.0, 0.0);
        if (o != p.centre) {
            const double scl = 1.0 / (double)cnt[r];
            out = mk(nan_to_num(acc_re[r] * scl), nan_to_num(acc_im[r] * scl));
        }
        gstore(p.recov + o, out);
    }
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

extern "C" int32_t scint_rev_map(const scint_c128* thth, const scint_c128* vec, const double* w,
                                 int32_t rank1, const double* th_cents, int64_t N,
                                 const scint_cs_geom* geom, double eta, int32_t hermitian,
                                 scint_c128* recov_out, void* stream_) {
    SCINT_REQUIRE(geom && th_cents && recov_out, "rev_map: null pointer");
    SCINT_REQUIRE(rank1 ? (vec && w) : (thth != nullptr), "rev_map: missing input");
    SCINT_REQUIRE(N >= 1, "rev_map: bad N");
    hipStream_t stream = (hipStream_t)stream_;
    const GeomDev g = to_dev(*geom);
    RevParams p;
    p.thth = (const cplx*)thth; p.ld = N;
    p.vec = (const cplx*)vec; p.w = w; p.rank1 = rank1;
    p.th = th_cents; p.N = (int)N;
    p.eta = eta; p.two_eta = 2 * eta;
    p.hermitian = hermitian;
    p.slab = (int)std::min<int64_t>(g.ntau, kRevSlab);
    // the bin the i == j terms fall in (fd_map = 0, tau_map = eta*0)
    const int64_t cbx = hist_bin(0.0, g.fd0, g.fd1_step, g.nfd);
    const int64_t cby = hist_bin(eta * 0.0, g.tau0, g.tau1_step, g.ntau);
    p.centre = (cbx >= 0 && cby >= 0) ? cby * g.nfd + cbx : -1;
    p.recov = (cplx*)recov_out;
    dim3 grid((unsigned)g.nfd, (unsigned)ceil_div(g.ntau, p.slab));
    SCINT_REQUIRE(grid.y <= 65535, "rev_map: ntau too large");
    hipLaunchKernelGGL(rev_gather_kernel, grid, dim3(256), (size_t)p.slab * 20, stream, p, g);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}
