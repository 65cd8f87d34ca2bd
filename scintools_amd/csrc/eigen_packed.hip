// eigen_packed.hip -- the eta sweep: packed gather + batched Hermitian Lanczos on the
// tile-packed theta-theta matrices (packed.hpp).  This is the headline path
// (scint_eval_sweep); eigen.hip keeps the full-matrix solver for user-supplied matrices.
//
// Per Lanczos step j, for all jobs of a batch at once:
//
//   pk_matvec_kernel  one workgroup per strip of <= 8 tiles (I, J0..J1): streams the tiles
//                     once (16 independent 1-KiB wave loads in flight per wave), forms
//                     the row-block partial  sum_J A_IJ x_J  (64 lanes stride the columns,
//                     wave-shuffle reduction at the end of the strip) and, per off-diagonal
//                     tile, the column-block partial  A_IJ^H x_I  (lane-local over rows,
//                     4-wave LDS reduction every 4 tiles; x_I is broadcast with v_readlane).
//                     The loop is software-pipelined over half tiles so that >= 8 KiB per
//                     wave is always in flight.  x = q_j is never stored normalised: every
//                     workgroup rebuilds x = (u_{j-1} - alpha_{j-1} q_{j-1}) / beta_{j-1}
//                     from the previous step's vectors and partial dot products.
//                     HBM bound: 8 N^2 bytes per job-step (algorithmic = actual).
//   pk_reduce_kernel  per 64-row block: fixed-order sum of its row/column partials,
//                     u_j = A q_j - beta_{j-1} q_{j-1}, q_j, and the partials of
//                     alpha_j = q_j^H u_j and |u_j|^2  (beta_j^2 = |u_j|^2 - alpha_j^2).
//   pk_check_kernel   (every 4 steps) top two Ritz values of T_k by 64-lane multisection on
//                     the Sturm count, Ritz residual by the backward recurrence, and the
//                     a-posteriori bound  err <= min(resid, resid^2 / (theta_1 - theta_2)).
//
// Scheduling: the `batch` slots are split into two groups driven on two streams: while the host
// waits for / reads back / refills one group, the other group's kernels keep the GPU busy.
// Within a group the slots are kept full -- as soon as a curvature converges (flags are read
// back every 4 launches) its slot is re-filled with the next eta of the sweep: gather + init
// for the new slots only, then the common step launches continue.  Every job carries the launch
// index it started at, so jobs at different Lanczos steps share one launch (continuous
// batching); per-job arithmetic does not depend on the schedule.
//
// Stopping: err <= tol * |theta_1| (tol = 1e-12 by default, i.e. 1000x tighter than the
// 1e-9 parity target against ARPACK) and the Ritz value moved by < 1e3 tol |theta_1| over the
// last 4 steps.  No atomics anywhere: results are bit-reproducible and independent of how the
// etas are batched.
#include <math.h>

#include <algorithm>
#include <map>
#include <vector>

#include "packed.hpp"
#include "prof.hpp"

namespace scint {

constexpr int kCheckEvery = 4;
constexpr int kFirstCheck = 8;
constexpr int kMaxK = 512;   // upper bound on Lanczos steps held in LDS by the check kernel

struct StepScalars { double alpha, beta, inv; };

// Element (row, col) of the packed Hermitian matrix, row != col blocks handled by symmetry.
__device__ inline cplx packed_at(const PackedJob& jb, int r, int c) {
    const int br = r / kTB, bc = c / kTB;
    if (bc >= br) return jb.tiles[(tile_offset(jb.nb, br) + (bc - br)) * kTileElems + (r % kTB) * kTB + (c % kTB)];
    return conj(jb.tiles[(tile_offset(jb.nb, bc) + (br - bc)) * kTileElems + (c % kTB) * kTB + (r % kTB)]);
}

// u_{-1} := v0 = row n/2 of theta-theta (Eval_calc, ththmod.py:398), q_{-1} := 0
__global__ void __launch_bounds__(64) pk_init_kernel(const PackedJob* jobs, const int32_t* slots) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K == 0 && e == 0) { jb.state[0] = 0; jb.state[1] = 0; jb.result[3] = -INFINITY; }
    if (K >= jb.nb) return;
    const int r = K * kTB + e;
    cplx v = mk(0.0, 0.0);
    if (r < jb.n && jb.n >= 2) v = packed_at(jb, jb.n / 2, r);
    jb.U[0][r] = v;
    jb.U[1][r] = mk(0.0, 0.0);
    jb.Q[(int64_t)(jb.qslots - 1) * jb.qstride + r] = mk(0.0, 0.0);   // "q_{-1}" = 0
    jb.Q[r] = mk(0.0, 0.0);
    const double p = wave_sum(norm2(v));
    if (e == 0) {
        jb.apart[0][K] = 0.0; jb.upart[0][K] = p;
        jb.apart[1][K] = 0.0; jb.upart[1][K] = 0.0;
    }
}

constexpr int kMaxStrip = 16;
constexpr int kFlush = 4;      // column partials are reduced across the 4 waves every kFlush tiles

// alpha_{j-1}, beta_{j-1} by ONE wavefront (no barriers): nb <= 64*k entries, fixed order
__device__ inline StepScalars step_scalars_wave(const double* __restrict__ ap, const double* __restrict__ up,
                                                int nb, int lane) {
    double a = 0.0, uu = 0.0;
    for (int i = lane; i < nb; i += 64) { a += gload(ap + i); uu += gload(up + i); }
    StepScalars s;
    s.alpha = wave_sum(a);
    uu = wave_sum(uu);
    const double b2 = uu - s.alpha * s.alpha;
    s.beta = b2 > 0.0 ? sqrt(b2) : 0.0;
    s.inv = s.beta > 0.0 ? 1.0 / s.beta : 0.0;
    return s;
}

// wave-uniform broadcast of lane `src`'s double through the scalar unit (no LDS, no VGPR)
__device__ inline double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256, 2)
pk_matvec_kernel(const PackedJob* __restrict__ jobs, const Strip* __restrict__ strips, int launch) {
    __shared__ cplx cred[4][kFlush][kTB];      // per-wave column partials of kFlush tiles (16 KiB)
    const Strip st = strips[blockIdx.x];
    const PackedJob* __restrict__ jp = jobs + st.job;
    const int step = launch - jp->start;
    if (jp->n < 2 || step < 0 || step >= jp->max_steps || jp->state[0]) return;
    const int par = step & 1;
    const int nb = jp->nb;
    const cplx* __restrict__ Up = par ? jp->U[1] : jp->U[0];
    const int qs = jp->qslots;
    const cplx* __restrict__ Qp = jp->Q + (int64_t)((step + qs - 1) % qs) * jp->qstride;   // q_{j-1}
    const cplx* __restrict__ tiles = jp->tiles;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int I = st.I;
    const int64_t t0 = tile_offset(nb, I);
    const int ntile = st.J1 - st.J0;
    // Software pipeline over half tiles (8 rows x 64 columns = 8 KiB per wave): the next half
    // tile's loads are always in flight while the current one is consumed.  The first loads go
    // out before anything else -- they do not depend on the step scalars.
    const cplx* __restrict__ tp = tiles + (t0 + (st.J0 - I)) * kTileElems + (16 * w) * kTB + lane;
    cplx a0[8], a1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a0[r] = gload_nt(tp + r * kTB);
    const StepScalars sc = step_scalars_wave(par ? jp->apart[1] : jp->apart[0],
                                             par ? jp->upart[1] : jp->upart[0], nb, lane);
    // lane l of every wave holds x_I[l]; rows read it back with v_readlane (scalar broadcast)
    cplx xIr;
    {
        const cplx u = gload(Up + I * kTB + lane), q = gload(Qp + I * kTB + lane);
        xIr = mk((u.x - sc.alpha * q.x) * sc.inv, (u.y - sc.alpha * q.y) * sc.inv);
    }
    cplx accR[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) accR[r] = mk(0.0, 0.0);
    cplx* __restrict__ colpart = jp->colpart;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const int J = st.J0 + t;
        const cplx* __restrict__ tc = tp + (int64_t)t * kTileElems;
#pragma unroll
        for (int r = 0; r < 8; ++r) a1[r] = gload_nt(tc + (8 + r) * kTB);   // second half of this tile
        const cplx uj = gload(Up + J * kTB + lane), qj = gload(Qp + J * kTB + lane);
        const cplx xJ = mk((uj.x - sc.alpha * qj.x) * sc.inv, (uj.y - sc.alpha * qj.y) * sc.inv);
        cplx c = mk(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            accR[r] = accR[r] + a0[r] * xJ;
            const cplx xi = mk(readlane_f64(xIr.x, 16 * w + r), readlane_f64(xIr.y, 16 * w + r));
            c = mk(c.x + a0[r].x * xi.x + a0[r].y * xi.y, c.y + a0[r].x * xi.y - a0[r].y * xi.x);   // conj(a) x_I
        }
        if (t + 1 < ntile) {
#pragma unroll
            for (int r = 0; r < 8; ++r) a0[r] = gload_nt(tc + kTileElems + r * kTB);   // first half of the next tile
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            accR[8 + r] = accR[8 + r] + a1[r] * xJ;
            const cplx xi = mk(readlane_f64(xIr.x, 16 * w + 8 + r), readlane_f64(xIr.y, 16 * w + 8 + r));
            c = mk(c.x + a1[r].x * xi.x + a1[r].y * xi.y, c.y + a1[r].x * xi.y - a1[r].y * xi.x);
        }
        cred[w][t & (kFlush - 1)][lane] = c;   // this wave's own slot
        if ((t & (kFlush - 1)) == kFlush - 1 || t + 1 == ntile) {
            // cross-wave reduction of the last <= kFlush tiles' column partials: wave w takes tile w
            __syncthreads();
            const int tb = t & ~(kFlush - 1);
            const int tt = tb + w;
            if (tt <= t) {
                const int Jt = st.J0 + tt;
                if (Jt != I) {
                    const cplx s = ((cred[0][w][lane] + cred[1][w][lane]) + cred[2][w][lane]) + cred[3][w][lane];
                    gstore(colpart + (t0 + (Jt - I)) * kTB + lane, s);
                }
            }
            __syncthreads();
        }
    }
    cplx* __restrict__ rowpart = jp->rowpart;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const cplx s = wave_sum(accR[r]);
        if (lane == 0) gstore(rowpart + (int64_t)st.index * kTB + 16 * w + r, s);
    }
}

constexpr int kRedGroups = 16;   // wavefronts per reduce block: each sums every 16th partial vector

__global__ void __launch_bounds__(64 * kRedGroups)
pk_reduce_kernel(const PackedJob* __restrict__ jobs, int launch) {
    __shared__ cplx part[kRedGroups][kTB];
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x;
    const int step = launch - jb.start;
    if (K >= jb.nb || jb.n < 2 || step < 0 || step >= jb.max_steps || jb.state[0]) return;
    const int par = step & 1;
    const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
    // fixed summation order: group g adds entries g, g+16, g+32, ... of the list
    // [row-strip partials of block row K, then column partials of tiles (0..K-1, K)];
    // the 16 group sums are then added in group order.  Few entries per group keeps the
    // dependent-load chain short.
    const int s0 = jb.row_strip0[K], nrow = jb.row_strip0[K + 1] - s0;
    cplx acc = mk(0.0, 0.0);
    for (int idx = g; idx < nrow + K; idx += kRedGroups) {
        if (idx < nrow) acc = acc + gload(jb.rowpart + (int64_t)(s0 + idx) * kTB + e);
        else {
            const int I = idx - nrow;
            acc = acc + gload(jb.colpart + (tile_offset(jb.nb, I) + (K - I)) * kTB + e);
        }
    }
    part[g][e] = acc;
    __syncthreads();
    if (g == 0) {
        // the same fixed-order scalar sums the mat-vec kernel evaluated
        const StepScalars sc = step_scalars_wave(par ? jb.apart[1] : jb.apart[0],
                                                 par ? jb.upart[1] : jb.upart[0], jb.nb, e);
        cplx total = part[0][e];
#pragma unroll
        for (int k = 1; k < kRedGroups; ++k) total = total + part[k][e];
        const int r = K * kTB + e;
        const cplx* __restrict__ Up = par ? jb.U[1] : jb.U[0];
        const cplx* __restrict__ Qp = jb.Q + (int64_t)((step + jb.qslots - 1) % jb.qslots) * jb.qstride;
        cplx* __restrict__ Un = par ? jb.U[0] : jb.U[1];
        cplx* __restrict__ Qn = jb.Q + (int64_t)(step % jb.qslots) * jb.qstride;
        const cplx up = gload(Up + r), qp = gload(Qp + r);
        const cplx qn = mk((up.x - sc.alpha * qp.x) * sc.inv, (up.y - sc.alpha * qp.y) * sc.inv);
        const cplx t = mk(total.x - sc.beta * qp.x, total.y - sc.beta * qp.y);
        gstore(Un + r, t);
        gstore(Qn + r, qn);
        const double pa = wave_sum(qn.x * t.x + qn.y * t.y);   // Re(conj(q) u)
        const double pu = wave_sum(norm2(t));
        if (e == 0) {
            (par ? jb.apart[0] : jb.apart[1])[K] = pa;
            (par ? jb.upart[0] : jb.upart[1])[K] = pu;
            if (K == 0) {
                if (step > 0) jb.alpha[step - 1] = sc.alpha;
                jb.beta[step] = sc.beta;
            }
        }
    }
}

// eigenvalues of T_k (diag a[0..k), off-diagonal b[1..k)) strictly below x
__device__ inline int sturm_count(const double* a, const double* b, int k, double x, double tiny) {
    int cnt = 0;
    double d = a[0] - x;
    if (fabs(d) < tiny) d = -tiny;
    cnt += d < 0.0;
    for (int i = 1; i < k; ++i) {
        d = (a[i] - x) - b[i] * b[i] / d;
        if (fabs(d) < tiny) d = -tiny;
        cnt += d < 0.0;
    }
    return cnt;
}

// smallest x in (lo, hi] with count(x) >= target, by 64-lane multisection; requires
// count(lo) < target <= count(hi).  Returns the midpoint of the final bracket.
__device__ inline double multisect(const double* a, const double* b, int k, int target, double lo,
                                   double hi, double tiny, int lane) {
    for (int round = 0; round < 48; ++round) {
        const double wdt = hi - lo;
        if (!(wdt > 0.0)) break;
        const double x = lo + wdt * ((double)(lane + 1) / 65.0);
        const int ok = (x > lo && x < hi) ? (sturm_count(a, b, k, x, tiny) >= target) : 0;
        const unsigned long long m = __ballot(ok);
        double nlo, nhi;
        if (m == 0ull) { nlo = __shfl(x, 63, 64); nhi = hi; }
        else {
            const int first = __ffsll((long long)m) - 1;
            nhi = __shfl(x, first, 64);
            nlo = first > 0 ? __shfl(x, first - 1, 64) : lo;
        }
        if (!(nlo > lo) && !(nhi < hi)) break;
        if (nlo > lo) lo = nlo;
        if (nhi < hi) hi = nhi;
        if (hi - lo <= 2e-16 * fmax(fabs(lo), fabs(hi))) break;
    }
    return 0.5 * (lo + hi);
}

__global__ void __launch_bounds__(64) pk_check_kernel(const PackedJob* jobs, int launches_done) {
    __shared__ double a[kMaxK + 1];
    __shared__ double b[kMaxK + 2];
    const PackedJob jb = jobs[blockIdx.x];
    if (jb.state[0]) return;
    const int lane = threadIdx.x;
    const int k_done = launches_done - jb.start;   // Lanczos steps this job has completed
    if (jb.n < 2) {
        if (lane == 0) {
            jb.state[0] = 1;
            jb.status_out[0] = SCINT_E_EMPTY;
            jb.eig_out[0] = nan("");
            if (jb.iters_out) jb.iters_out[0] = 0;
        }
        return;
    }
    if (k_done < kFirstCheck && k_done < jb.max_steps) return;
    const int k = min(k_done, jb.max_steps);
    // alpha_{k-1}, beta_{k-1} are still in the partials of the last reduce kernel
    const int par = k & 1;
    double al = 0.0, uu = 0.0;
    const double* __restrict__ ap = par ? jb.apart[1] : jb.apart[0];
    const double* __restrict__ upp = par ? jb.upart[1] : jb.upart[0];
    for (int i = lane; i < jb.nb; i += 64) { al += ap[i]; uu += upp[i]; }
    al = wave_sum(al);
    uu = wave_sum(uu);
    const double b2 = uu - al * al;
    const double beta_k = b2 > 0.0 ? sqrt(b2) : 0.0;
    for (int i = lane; i < k - 1; i += 64) a[i] = jb.alpha[i];
    for (int i = lane + 1; i < k; i += 64) b[i] = jb.beta[i];
    if (lane == 0) { a[k - 1] = al; b[0] = 0.0; }
    __syncthreads();

    double lo = INFINITY, hi = -INFINITY, scale = 0.0;
    for (int i = lane; i < k; i += 64) {
        const double off = (i > 0 ? fabs(b[i]) : 0.0) + (i + 1 < k ? fabs(b[i + 1]) : 0.0);
        lo = fmin(lo, a[i] - off);
        hi = fmax(hi, a[i] + off);
        scale = fmax(scale, fabs(a[i]) + off);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
        scale = fmax(scale, __shfl_xor(scale, o, 64));
    }
    const bool finite = isfinite(lo) && isfinite(hi) && isfinite(beta_k);
    double theta = nan(""), theta2 = -INFINITY, resid = nan(""), err = nan("");
    if (finite && scale == 0.0 && beta_k == 0.0) {
        // T_k = 0: an all-zero theta-theta (e.g. every delay masked); ARPACK returns 0 as well
        theta = 0.0; theta2 = 0.0; resid = 0.0; err = 0.0;
    } else if (finite) {
        const double tiny = scale * 1e-300 + 1e-300;
        lo = lo - 1e-15 * fabs(lo) - 1e-300;   // count(lo) == 0
        hi = hi + 1e-15 * fabs(hi) + 1e-300;   // count(hi) == k
        theta = multisect(a, b, k, k, lo, hi, tiny, lane);
        if (k >= 2) theta2 = multisect(a, b, k, k - 1, lo, theta, tiny, lane);
        // Ritz residual beta_k |s_{k-1}|, s = eigenvector of T_k by the backward recurrence
        if (lane == 0) {
            double* __restrict__ sv = jb.want_vec ? jb.svec : nullptr;
            double sk = 1.0, skp1 = 0.0, nrm = 1.0, last = 1.0;
            if (sv) sv[k - 1] = 1.0;
            for (int i = k - 1; i >= 1; --i) {
                const double bi = b[i];
                double sm1 = (bi != 0.0) ? ((theta - a[i]) * sk - (i + 1 < k ? b[i + 1] * skp1 : 0.0)) / bi : 0.0;
                if (!isfinite(sm1)) sm1 = 0.0;
                if (fabs(sm1) > 1e150) {
                    const double f = 1e-150;
                    sm1 *= f; sk *= f; last *= f; nrm *= f * f;
                    if (sv) for (int t = i; t < k; ++t) sv[t] *= f;
                }
                if (sv) sv[i - 1] = sm1;
                nrm += sm1 * sm1;
                skp1 = sk;
                sk = sm1;
            }
            const double inv = 1.0 / sqrt(nrm);
            if (sv) for (int t = 0; t < k; ++t) sv[t] *= inv;
            resid = beta_k * fabs(last) * inv;
        }
        resid = __shfl(resid, 0, 64);
        const double gap = theta - theta2;
        err = (gap > resid) ? resid * resid / gap : resid;
    }
    if (lane == 0) {
        const double prev = jb.result[3];
        const double at = fmax(fabs(theta), 1e-300);
        const bool settled = (theta - prev) <= 1e3 * jb.tol * at;
        const bool exact = finite && (k >= jb.n || beta_k == 0.0);
        // eigenvalue only: a-posteriori bound on theta; eigenvector wanted: the Ritz residual
        // itself (vector error ~ resid / gap)
        const bool ok = jb.want_vec ? (resid <= jb.tol * at) : (err <= jb.tol * at && settled);
        const bool conv = finite && (ok || exact);
        const bool stop = conv || !finite || k >= jb.max_steps;
        jb.result[0] = theta; jb.result[1] = err; jb.result[2] = resid; jb.result[3] = theta;
        if (stop) {
            jb.state[0] = 1;
            jb.state[1] = k;
            jb.eig_out[0] = jb.want_vec ? theta : fabs(theta);   // modeler keeps the sign of w
            if (jb.iters_out) jb.iters_out[0] = k;
            jb.status_out[0] = (!finite || !isfinite(theta)) ? SCINT_E_NONFINITE : (conv ? SCINT_OK : SCINT_E_NOCONV);
        }
    }
}

// Ritz vector of finished jobs: y = sum_j s_j q_j, then normalised (fixed-order reductions).
// grid (nb, njobs); vec_out rows are `vstride` apart and indexed by the job's eta index.
__global__ void __launch_bounds__(64) pk_ritz_kernel(const PackedJob* jobs, const int32_t* slots,
                                                     const int64_t* eta_index, cplx* vec_out,
                                                     int64_t vstride) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K >= jb.nb) return;
    const int k = jb.state[1];
    const int r = K * kTB + e;
    cplx y = mk(0.0, 0.0);
    for (int j = 0; j < k; ++j) {
        const cplx q = jb.Q[(int64_t)j * jb.qstride + r];
        const double s = jb.svec[j];
        y = mk(y.x + s * q.x, y.y + s * q.y);
    }
    cplx* out = vec_out + eta_index[blockIdx.y] * vstride;
    if (r < jb.n) out[r] = y;
    const double p = wave_sum(r < jb.n ? norm2(y) : 0.0);
    if (e == 0) jb.upart[0][K] = p;   // the job is finished: its partial arrays are free
}

__global__ void __launch_bounds__(64) pk_ritz_scale_kernel(const PackedJob* jobs, const int32_t* slots,
                                                           const int64_t* eta_index, cplx* vec_out,
                                                           int64_t vstride) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K >= jb.nb) return;
    double t = 0.0;
    for (int i = e; i < jb.nb; i += 64) t += jb.upart[0][i];
    t = wave_sum(t);
    const double inv = t > 0.0 ? 1.0 / sqrt(t) : 0.0;
    const int r = K * kTB + e;
    cplx* out = vec_out + eta_index[blockIdx.y] * vstride;
    if (r < jb.n) out[r] = mk(out[r].x * inv, out[r].y * inv);
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------
struct SlabLayout {
    size_t tiles, U0, U1, Q, svec, rowpart, colpart, row_strip0, apart0, apart1, upart0, upart1,
        alpha, beta, result, total;
    int qslots;
};

static int max_strips(int nb) {
    const int S = strip_len_for(nb);
    int n = 0;
    for (int I = 0; I < nb; ++I) n += strips_in_row(nb, I, S);
    return n;
}

static SlabLayout slab_layout(int nbmax, int max_steps, bool want_vec) {
    SlabLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    // the strip count is not monotone in nb across the strip-length thresholds: take the max
    int smax = 0;
    for (int nb = 1; nb <= nbmax; ++nb) smax = std::max(smax, max_strips(nb));
    L.tiles = take(sizeof(cplx) * (size_t)tile_count(nbmax) * kTileElems);
    L.U0 = take(sizeof(cplx) * (size_t)nbmax * kTB);
    L.U1 = take(sizeof(cplx) * (size_t)nbmax * kTB);
    L.qslots = want_vec ? max_steps + 1 : 2;
    L.Q = take(sizeof(cplx) * (size_t)nbmax * kTB * (size_t)L.qslots);
    L.svec = take(sizeof(double) * (size_t)(max_steps + 2));
    L.rowpart = take(sizeof(cplx) * (size_t)smax * kTB);
    L.colpart = take(sizeof(cplx) * (size_t)tile_count(nbmax) * kTB);
    L.row_strip0 = take(sizeof(int32_t) * (size_t)(nbmax + 1));
    L.apart0 = take(sizeof(double) * (size_t)nbmax);
    L.apart1 = take(sizeof(double) * (size_t)nbmax);
    L.upart0 = take(sizeof(double) * (size_t)nbmax);
    L.upart1 = take(sizeof(double) * (size_t)nbmax);
    L.alpha = take(sizeof(double) * (size_t)(max_steps + 2));
    L.beta = take(sizeof(double) * (size_t)(max_steps + 3));
    L.result = take(sizeof(double) * 4);
    L.total = align_up(off, 256);
    return L;
}

struct BatchLayout {
    SlabLayout slab;
    int smax;
    size_t jobs, strips, states, slots, fin_slots, fin_eta, geoms, total;
};

static BatchLayout batch_layout(int nbmax, int max_steps, int nbatch, bool want_vec, int64_t ncs) {
    BatchLayout B;
    B.slab = slab_layout(nbmax, max_steps, want_vec);
    B.smax = 0;
    for (int nb = 1; nb <= nbmax; ++nb) B.smax = std::max(B.smax, max_strips(nb));
    size_t off = B.slab.total * (size_t)nbatch;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    B.jobs = take(sizeof(PackedJob) * (size_t)nbatch);
    B.strips = take(sizeof(Strip) * (size_t)nbatch * (size_t)B.smax);
    B.states = take(sizeof(int32_t) * 4 * (size_t)nbatch);
    B.slots = take(sizeof(int32_t) * (size_t)nbatch);
    B.fin_slots = take(sizeof(int32_t) * (size_t)nbatch);
    B.fin_eta = take(sizeof(int64_t) * (size_t)nbatch);
    B.geoms = take(sizeof(GeomDev) * (size_t)ncs);
    B.total = align_up(off, 256);
    return B;
}


static int32_t sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch, int32_t max_iter,
                                     bool want_vec, int64_t ncs, size_t* bytes) {
    SCINT_REQUIRE(bytes && M >= 1 && neta >= 1 && batch >= 1 && max_iter >= 1 && ncs >= 1,
                  "sweep_workspace_bytes: bad arguments");
    const int nbmax = (int)ceil_div(M, kTB);
    const int steps = (int)std::min<int64_t>(std::min<int64_t>(max_iter, M), kMaxK);
    const int nbatch = (int)std::min(batch, neta);
    *bytes = batch_layout(nbmax, steps, nbatch, want_vec, ncs).total + 4096;
    return SCINT_OK;
}

// Pinned read-back buffer for the per-slot state words, kept per host thread and grown on
// demand (a hipHostMalloc per sweep call costs more than a small sweep).
static int32_t* pinned_flags(size_t count) {
    thread_local int32_t* buf = nullptr;
    thread_local size_t cap = 0;
    if (count > cap) {
        if (buf) (void)hipHostFree(buf);
        buf = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(count, 1024);
        if (hipHostMalloc(&buf, sizeof(int32_t) * want) != hipSuccess) {
            set_error("scint: hipHostMalloc of the flag buffer failed");
            buf = nullptr;
            return nullptr;
        }
        cap = want;
    }
    return buf;
}

// A second stream per host thread: the sweep alternates two groups of slots so that while the
// host reads one group's convergence flags and refills its slots, the other group's kernels
// keep the GPU busy (no sync bubbles, and the latency-bound reduce/check kernels of one group
// overlap the bandwidth-bound mat-vec of the other).
static hipStream_t second_stream() {
    thread_local std::map<int, hipStream_t> streams;      // one per (host thread, device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto it = streams.find(dev);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    streams[dev] = s;
    return s;
}
static hipEvent_t make_event() {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    return e;
}

struct SweepShared {
    // problem
    const cplx* cs; int64_t cs_stride; const int32_t* cs_index; const double* th_cents; int64_t M;
    const int32_t* keep_idx; const int32_t* keep_n; const double* etas; int64_t neta;
    double* eigs_out; int32_t* status_out; int32_t* iters_out;
    bool want_vec; cplx* vec_out; int64_t vstride;
    int nbmax, steps_cap;
    // device tables
    char* base; const SlabLayout* L; PackedJob* jobs_dev; const GeomDev* geoms_dev;
    // the queue of curvatures still to be started
    int64_t next_eta;
};

// One group of slots [s0, s1) driven on its own stream.
struct SweepGroup {
    SweepShared* sh;
    int s0, s1;
    hipStream_t stream;
    hipEvent_t done;
    Strip* strips_dev; int32_t* states_dev; int32_t* slots_dev; int32_t* fin_slots_dev; int64_t* fin_eta_dev;
    int32_t* flags;                              // pinned, 4 words per slot of the group
    std::vector<PackedJob>* jobs;                // host copy of the whole job table
    std::vector<int64_t> slot_eta;               // per slot of the group: running eta or -1
    std::vector<Strip> strips;
    std::vector<int32_t> rs_all, fresh, fin_slots;
    std::vector<int64_t> fin_eta;
    int launch = 0, active = 0;
    bool in_flight = false;

    int nslots() const { return s1 - s0; }

    // refill idle slots, then enqueue kCheckEvery steps + the check + the flag read-back
    int32_t enqueue() {
        SweepShared& S = *sh;
        const SlabLayout& L = *S.L;
        hipError_t he = hipSuccess;
        fresh.clear();
        for (int s = s0; s < s1 && S.next_eta < S.neta; ++s) {
            if (slot_eta[(size_t)(s - s0)] >= 0) continue;
            const int64_t e = S.next_eta++;
            slot_eta[(size_t)(s - s0)] = e;
            ++active;
            PackedJob& J = (*jobs)[(size_t)s];
            const int n = S.keep_n[e];
            J.eta = S.etas[e]; J.two_eta = 2 * S.etas[e];
            const int64_t c = S.cs_index ? S.cs_index[e] : 0;
            J.cs = S.cs + c * S.cs_stride; J.th = S.th_cents + c * S.M; J.geom = (int32_t)c;
            J.keep = S.keep_idx + e * S.M; J.n = n; J.nb = (int)ceil_div(std::max(n, 1), kTB);
            J.max_steps = std::min(S.steps_cap, std::max(n, 1));
            J.strip_len = strip_len_for(J.nb);
            J.start = launch;
            J.eig_out = S.eigs_out + e; J.status_out = S.status_out + e;
            J.iters_out = S.iters_out ? S.iters_out + e : nullptr;
            fresh.push_back(s);
        }
        if (active == 0) { in_flight = false; return SCINT_OK; }
        if (!fresh.empty()) {
            // strips of every running job of the group, block row by block row
            strips.clear();
            int nb_fresh = 0;
            for (int s = s0; s < s1; ++s) {
                if (slot_eta[(size_t)(s - s0)] < 0) continue;
                const PackedJob& J = (*jobs)[(size_t)s];
                int32_t* rs0 = rs_all.data() + (size_t)(s - s0) * (size_t)(S.nbmax + 1);
                int idx = 0;
                for (int I = 0; I < J.nb; ++I) {
                    rs0[I] = idx;
                    for (int J0 = I; J0 < J.nb; J0 += J.strip_len) {
                        Strip st;
                        st.job = s; st.I = I; st.J0 = J0; st.J1 = std::min(J.nb, J0 + J.strip_len); st.index = idx++;
                        strips.push_back(st);
                    }
                }
                rs0[J.nb] = idx;
            }
            // longest strips first: the short ones of the last block rows then fill the tail
            // of the launch (dispatch order only; the arithmetic does not depend on it)
            std::stable_sort(strips.begin(), strips.end(), [](const Strip& a, const Strip& b) {
                return (a.J1 - a.J0) > (b.J1 - b.J0);
            });
            for (int s : fresh) {
                const PackedJob& J = (*jobs)[(size_t)s];
                nb_fresh = std::max(nb_fresh, J.nb);
                he = hipMemcpyAsync(S.base + L.total * (size_t)s + L.row_strip0,
                                    rs_all.data() + (size_t)(s - s0) * (size_t)(S.nbmax + 1),
                                    sizeof(int32_t) * (size_t)(J.nb + 1), hipMemcpyHostToDevice, stream);
                if (he != hipSuccess) break;
            }
            if (he == hipSuccess)
                he = hipMemcpyAsync(S.jobs_dev + s0, jobs->data() + s0, sizeof(PackedJob) * (size_t)nslots(),
                                    hipMemcpyHostToDevice, stream);
            if (he == hipSuccess)
                he = hipMemcpyAsync(strips_dev, strips.data(), sizeof(Strip) * strips.size(), hipMemcpyHostToDevice, stream);
            if (he == hipSuccess)
                he = hipMemcpyAsync(slots_dev, fresh.data(), sizeof(int32_t) * fresh.size(), hipMemcpyHostToDevice, stream);
            if (he == hipSuccess) he = hipStreamSynchronize(stream);   // host vectors are reused; the stream is idle here
            if (he != hipSuccess) return hip_fail(he, "sweep job upload", __FILE__, __LINE__);
            int32_t rc = launch_gather_packed(S.geoms_dev, S.M, S.jobs_dev, slots_dev, (int)fresh.size(), nb_fresh, stream);
            if (rc != SCINT_OK) return rc;
            hipLaunchKernelGGL(pk_init_kernel, dim3((unsigned)nb_fresh, (unsigned)fresh.size()), dim3(64), 0, stream,
                               S.jobs_dev, slots_dev);
        }
        int nb_run = 1;
        for (int s = s0; s < s1; ++s)
            if (slot_eta[(size_t)(s - s0)] >= 0) nb_run = std::max(nb_run, (*jobs)[(size_t)s].nb);
        const unsigned nstrips = (unsigned)strips.size();
        for (int i = 0; i < kCheckEvery; ++i, ++launch) {
            const int slot = profiler().begin(kProfMatvec, stream);
            hipLaunchKernelGGL(pk_matvec_kernel, dim3(nstrips), dim3(256), 0, stream, S.jobs_dev, strips_dev, launch);
            profiler().end(kProfMatvec, slot, stream);
            hipLaunchKernelGGL(pk_reduce_kernel, dim3((unsigned)nb_run, (unsigned)nslots()), dim3(64 * kRedGroups), 0,
                               stream, S.jobs_dev + s0, launch);
        }
        hipLaunchKernelGGL(pk_check_kernel, dim3((unsigned)nslots()), dim3(64), 0, stream, S.jobs_dev + s0, launch);
        he = hipGetLastError();
        if (he == hipSuccess)
            he = hipMemcpyAsync(flags, states_dev, sizeof(int32_t) * 4 * (size_t)nslots(), hipMemcpyDeviceToHost, stream);
        if (he == hipSuccess) he = hipEventRecord(done, stream);
        if (he != hipSuccess) return hip_fail(he, "sweep step", __FILE__, __LINE__);
        in_flight = true;
        return SCINT_OK;
    }

    // wait for the group's last chunk, retire converged curvatures (and export their vectors)
    int32_t harvest() {
        if (!in_flight) return SCINT_OK;
        SweepShared& S = *sh;
        hipError_t he = hipEventSynchronize(done);
        if (he != hipSuccess) return hip_fail(he, "sweep wait", __FILE__, __LINE__);
        in_flight = false;
        if (profiler().enabled) profiler().collect();
        fin_slots.clear();
        fin_eta.clear();
        int nb_fin = 1;
        for (int s = s0; s < s1; ++s) {
            const size_t k = (size_t)(s - s0);
            if (slot_eta[k] >= 0 && flags[4 * k] != 0) {
                if (S.want_vec && (*jobs)[(size_t)s].n >= 2) {
                    fin_slots.push_back(s);
                    fin_eta.push_back(slot_eta[k]);
                    nb_fin = std::max(nb_fin, (*jobs)[(size_t)s].nb);
                }
                slot_eta[k] = -1;             // finished: results were written by the check kernel
                (*jobs)[(size_t)s].n = 0;     // an idle slot's kernels exit at once (host copy only)
                --active;
            }
        }
        if (!fin_slots.empty()) {
            // export the Ritz vectors before the slots are re-used (the device job table still
            // describes the finished jobs: it is only rewritten at the next refill)
            he = hipMemcpyAsync(fin_slots_dev, fin_slots.data(), sizeof(int32_t) * fin_slots.size(),
                                hipMemcpyHostToDevice, stream);
            if (he == hipSuccess)
                he = hipMemcpyAsync(fin_eta_dev, fin_eta.data(), sizeof(int64_t) * fin_eta.size(),
                                    hipMemcpyHostToDevice, stream);
            if (he == hipSuccess) {
                const dim3 grid((unsigned)nb_fin, (unsigned)fin_slots.size());
                hipLaunchKernelGGL(pk_ritz_kernel, grid, dim3(64), 0, stream, S.jobs_dev, fin_slots_dev, fin_eta_dev,
                                   S.vec_out, S.vstride);
                hipLaunchKernelGGL(pk_ritz_scale_kernel, grid, dim3(64), 0, stream, S.jobs_dev, fin_slots_dev,
                                   fin_eta_dev, S.vec_out, S.vstride);
                he = hipGetLastError();
            }
            if (he == hipSuccess) he = hipStreamSynchronize(stream);
            if (he != hipSuccess) return hip_fail(he, "sweep ritz vectors", __FILE__, __LINE__);
        }
        return SCINT_OK;
    }
};

// Shared driver of scint_eval_sweep (eigenvalues), scint_eigvec_sweep (eigenpairs) and
// scint_eval_sweep_multi.  `ncs` conjugate spectra of one shape live `cs_stride` elements apart
// from `cs`, each with its own geometry geom[c] and theta grid th_cents + c*M; curvature e reads
// spectrum cs_index[e] (nullptr: all read spectrum 0).
static int32_t run_sweep(const scint_c128* cs, int64_t ncs, int64_t cs_stride, const int32_t* cs_index,
                         const scint_cs_geom* geom, const double* th_cents,
                         int64_t M, const int32_t* keep_idx, const int32_t* keep_n, const double* etas,
                         int64_t neta, double tol, int32_t max_iter, int64_t batch, double* eigs_out,
                         int32_t* status_out, int32_t* iters_out, bool want_vec, cplx* vec_out,
                         int64_t vstride, void* workspace, size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(cs && geom && th_cents && keep_idx && keep_n && etas && eigs_out && status_out && workspace,
                  "sweep: null pointer");
    SCINT_REQUIRE(M >= 1 && neta >= 1 && batch >= 1 && max_iter >= 1 && tol > 0, "sweep: bad arguments");
    SCINT_REQUIRE(ncs >= 1 && (ncs == 1 || cs_index), "sweep: bad spectrum table");
    for (int64_t c = 0; c < ncs; ++c) {
        SCINT_REQUIRE(geom[c].dtau > 0 && geom[c].dfd > 0, "sweep: tau and fd must be increasing");
        SCINT_REQUIRE(geom[c].ntau == geom[0].ntau && geom[c].nfd == geom[0].nfd,
                      "sweep: all conjugate spectra must have one shape");
    }
    SCINT_REQUIRE(!want_vec || (vec_out && vstride >= M), "sweep: bad eigenvector output");
    hipStream_t stream = (hipStream_t)stream_;
    size_t need = 0;
    sweep_workspace_bytes(M, neta, batch, max_iter, want_vec, ncs, &need);
    if (workspace_bytes < need) { set_error("scint: sweep workspace too small"); return SCINT_E_WORKSPACE; }
    const int nbmax = (int)ceil_div(M, kTB);
    const int steps_cap = (int)std::min<int64_t>(std::min<int64_t>(max_iter, M), kMaxK);
    const int nslots = (int)std::min(batch, neta);
    const BatchLayout BL = batch_layout(nbmax, steps_cap, nslots, want_vec, ncs);
    const SlabLayout& L = BL.slab;
    char* base = (char*)workspace;
    PackedJob* jobs_dev = (PackedJob*)(base + BL.jobs);
    Strip* strips_dev = (Strip*)(base + BL.strips);
    int32_t* states_dev = (int32_t*)(base + BL.states);
    int32_t* slots_dev = (int32_t*)(base + BL.slots);
    int32_t* fin_slots_dev = (int32_t*)(base + BL.fin_slots);
    int64_t* fin_eta_dev = (int64_t*)(base + BL.fin_eta);
    GeomDev* geoms_dev = (GeomDev*)(base + BL.geoms);
    std::vector<GeomDev> geoms_host((size_t)ncs);
    for (int64_t c = 0; c < ncs; ++c) geoms_host[(size_t)c] = to_dev(geom[c]);
    SCINT_HIP(hipMemcpyAsync(geoms_dev, geoms_host.data(), sizeof(GeomDev) * (size_t)ncs, hipMemcpyHostToDevice,
                             stream));
    SCINT_HIP(hipStreamSynchronize(stream));   // also: everything queued before us (the CS) is complete

    std::vector<PackedJob> jobs((size_t)nslots);
    int32_t* flags = pinned_flags((size_t)nslots * 4);
    if (!flags) return SCINT_E_HIP;
    for (int s = 0; s < nslots; ++s) {                    // static part of every slot
        char* sl = base + L.total * (size_t)s;
        PackedJob& J = jobs[(size_t)s];
        J.tiles = (cplx*)(sl + L.tiles);
        J.U[0] = (cplx*)(sl + L.U0); J.U[1] = (cplx*)(sl + L.U1);
        J.Q = (cplx*)(sl + L.Q); J.qstride = (int64_t)nbmax * kTB; J.qslots = L.qslots;
        J.want_vec = want_vec ? 1 : 0; J.svec = (double*)(sl + L.svec);
        J.rowpart = (cplx*)(sl + L.rowpart); J.colpart = (cplx*)(sl + L.colpart);
        J.row_strip0 = (const int32_t*)(sl + L.row_strip0);
        J.apart[0] = (double*)(sl + L.apart0); J.apart[1] = (double*)(sl + L.apart1);
        J.upart[0] = (double*)(sl + L.upart0); J.upart[1] = (double*)(sl + L.upart1);
        J.alpha = (double*)(sl + L.alpha); J.beta = (double*)(sl + L.beta);
        J.result = (double*)(sl + L.result); J.state = states_dev + 4 * s;
        J.tol = tol; J.pad0 = 0;
        J.n = 0; J.nb = 1; J.max_steps = 0; J.start = 0; J.strip_len = 1;
        J.eta = 0; J.two_eta = 0; J.keep = keep_idx;
        J.cs = (const cplx*)cs; J.th = th_cents; J.geom = 0; J.pad1 = 0;
        J.eig_out = eigs_out; J.status_out = status_out; J.iters_out = iters_out;
    }
    // idle slots must look idle on the device before any group launches over them
    SCINT_HIP(hipMemcpyAsync(jobs_dev, jobs.data(), sizeof(PackedJob) * (size_t)nslots, hipMemcpyHostToDevice, stream));
    SCINT_HIP(hipMemsetAsync(states_dev, 0, sizeof(int32_t) * 4 * (size_t)nslots, stream));
    SCINT_HIP(hipStreamSynchronize(stream));

    SweepShared S;
    S.cs = (const cplx*)cs; S.cs_stride = cs_stride; S.cs_index = cs_index; S.th_cents = th_cents; S.M = M;
    S.keep_idx = keep_idx; S.keep_n = keep_n; S.etas = etas; S.neta = neta;
    S.eigs_out = eigs_out; S.status_out = status_out; S.iters_out = iters_out;
    S.want_vec = want_vec; S.vec_out = vec_out; S.vstride = vstride;
    S.nbmax = nbmax; S.steps_cap = steps_cap;
    S.base = base; S.L = &L; S.jobs_dev = jobs_dev; S.geoms_dev = geoms_dev;
    S.next_eta = 0;

    // two groups on two streams when there are enough slots to split
    static const int forced_groups = [] { const char* e = getenv("SCINT_SWEEP_GROUPS"); return e ? atoi(e) : 0; }();
    hipStream_t s2 = second_stream();
    int ngroups = (nslots >= 4 && s2) ? 2 : 1;
    if (forced_groups == 1) ngroups = 1;
    SweepGroup G[2];
    int32_t rc = SCINT_OK;
    for (int g = 0; g < ngroups; ++g) {
        SweepGroup& q = G[g];
        q.sh = &S;
        q.s0 = g == 0 ? 0 : nslots / 2;
        q.s1 = (g == ngroups - 1) ? nslots : nslots / 2;
        q.stream = g == 0 ? stream : s2;
        q.done = make_event();
        if (!q.done) { set_error("scint: hipEventCreate failed"); rc = SCINT_E_HIP; }
        q.strips_dev = strips_dev + (size_t)q.s0 * (size_t)BL.smax;
        q.states_dev = states_dev + 4 * q.s0;
        q.slots_dev = slots_dev + q.s0;
        q.fin_slots_dev = fin_slots_dev + q.s0;
        q.fin_eta_dev = fin_eta_dev + q.s0;
        q.flags = flags + 4 * q.s0;
        q.jobs = &jobs;
        q.slot_eta.assign((size_t)q.nslots(), -1);
        q.rs_all.assign((size_t)q.nslots() * (size_t)(nbmax + 1), 0);
    }
    // pipeline: enqueue A, enqueue B, then repeatedly {harvest g, enqueue g} alternating groups
    for (int g = 0; g < ngroups && rc == SCINT_OK; ++g) rc = G[g].enqueue();
    while (rc == SCINT_OK) {
        bool any = false;
        for (int g = 0; g < ngroups && rc == SCINT_OK; ++g) {
            if (!G[g].in_flight) continue;
            any = true;
            rc = G[g].harvest();
            if (rc == SCINT_OK) rc = G[g].enqueue();
        }
        if (!any) break;
    }
    // leave nothing running on the internal stream, and order the caller's stream after it
    for (int g = 0; g < ngroups; ++g) {
        if (G[g].stream) (void)hipStreamSynchronize(G[g].stream);
        if (G[g].done) (void)hipEventDestroy(G[g].done);
    }
    return rc;
}

}  // namespace scint

using namespace scint;

extern "C" int32_t scint_eval_sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                                    int32_t max_iter, size_t* bytes) {
    return sweep_workspace_bytes(M, neta, batch, max_iter, false, 1, bytes);
}

extern "C" int32_t scint_eval_sweep(const scint_c128* cs, const scint_cs_geom* geom,
                                    const double* th_cents, int64_t M, const int32_t* keep_idx,
                                    const int32_t* keep_n, const double* etas, int64_t neta,
                                    double tol, int32_t max_iter, int64_t batch, double* eigs_out,
                                    int32_t* status_out, int32_t* iters_out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return run_sweep(cs, 1, 0, nullptr, geom, th_cents, M, keep_idx, keep_n, etas, neta, tol, max_iter, batch,
                     eigs_out, status_out, iters_out, false, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int32_t scint_eval_sweep_multi_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                                          int32_t max_iter, int64_t ncs, size_t* bytes) {
    return sweep_workspace_bytes(M, neta, batch, max_iter, false, ncs, bytes);
}

extern "C" int32_t scint_eval_sweep_multi(const scint_c128* cs_stack, int64_t ncs, int64_t cs_stride,
                                          const int32_t* cs_index, const scint_cs_geom* geoms,
                                          const double* th_stack, int64_t M, const int32_t* keep_idx,
                                          const int32_t* keep_n, const double* etas, int64_t neta,
                                          double tol, int32_t max_iter, int64_t batch, double* eigs_out,
                                          int32_t* status_out, int32_t* iters_out, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    SCINT_REQUIRE(cs_index != nullptr || ncs == 1, "eval_sweep_multi: cs_index required");
    return run_sweep(cs_stack, ncs, cs_stride, cs_index, geoms, th_stack, M, keep_idx, keep_n, etas, neta, tol,
                     max_iter, batch, eigs_out, status_out, iters_out, false, nullptr, 0, workspace,
                     workspace_bytes, stream);
}

extern "C" int32_t scint_eigvec_sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                                      int32_t max_iter, size_t* bytes) {
    return sweep_workspace_bytes(M, neta, batch, max_iter, true, 1, bytes);
}

extern "C" int32_t scint_eigvec_sweep(const scint_c128* cs, const scint_cs_geom* geom,
                                      const double* th_cents, int64_t M, const int32_t* keep_idx,
                                      const int32_t* keep_n, const double* etas, int64_t neta,
                                      double tol, int32_t max_iter, int64_t batch, double* w_out,
                                      scint_c128* vec_out, int64_t vec_stride, int32_t* status_out,
                                      int32_t* iters_out, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    return run_sweep(cs, 1, 0, nullptr, geom, th_cents, M, keep_idx, keep_n, etas, neta, tol, max_iter, batch,
                     w_out, status_out, iters_out, true, (cplx*)vec_out, vec_stride, workspace, workspace_bytes,
                     stream);
}
