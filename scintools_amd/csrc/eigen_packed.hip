// eigen_packed.hip -- the eta sweep: packed gather + batched two-vector (block) Hermitian Lanczos on
// the tile-packed theta-theta matrices (packed.hpp).  This is the headline path
// (scint_eval_sweep); eigen.hip keeps the full-matrix solver for user-supplied matrices.
//
// Per Lanczos pass j, for all jobs of a batch at once (details at the kernels):
//
//   pk2_coef_kernel    the step's 2x2 coefficients A_{j-1}, B_{j-1} from the previous step's fixed-order partial
//                      sums and the block Q_j they define, once per job (1024 rows per workgroup).
//   pk2_matvec_kernel  one workgroup per strip of <= 16 column tiles of TWO block rows, described by ONE record:
//                      streams the tiles once (16 independent 1-KiB wave loads in flight per wave) and
//                      multiplies BOTH vectors of the block Q_j: a wave owns a 16-column slice of every tile, so
//                      the column partials A_IJ^H X_I finish inside the wave (shuffles) and leave in one burst
//                      per strip; row partials sum_J A_IJ X_J by shuffles + one LDS step.  HBM bound:
//                      8 N (N + 1) bytes per pass.
//   pk2_reduce_kernel  per 64-row block: fixed-order sum of its row/column partials -> W_j and the
//                      partial sums of A_j = Q_j^H W_j and of the Gram matrix W_j^H W_j.
//   pk2_check_kernel   (every 2 passes, on the group's check stream) top two eigenvalues of the pentadiagonal T_k by 64-lane
//                      multisection on a banded LDL^H Sturm count, Ritz residual by inverse iteration,
//                      a-posteriori bound  err <= min(resid, resid^2 / (theta_1 - theta_2)).
//
// Mixed-precision eigenvalue sweep (opt-in, scint_sweep_precision; "Mixed precision" further down): the passes above run on a
// complex64 copy of the tiles (matvec32.hpp: pk2_matvec32_kernel, pk2_matvec_mixed_kernel), the check hands the two top Ritz
// vectors over (pk2_restart_kernel) to a certificate run on the complex128 tiles whose first residual is formed row by row
// (pk2_cert_resid_kernel); the value returned is that run's Ritz value under the same stopping rule.
//
// Scheduling (SweepGroup below): the `batch` slots are kept full -- when a curvature converges its
// slot is re-filled with the next eta of the sweep (continuous batching; every job carries the
// launch index it started at, so jobs at different Lanczos steps share one launch).  Two groups of
// slots run on two streams and fill each other's gaps; in each, the passes are queued in chunks of
// 3 + a convergence check, two chunks ahead of the state the host has seen, so a stream never
// waits for the host.  Per-job arithmetic does not depend on the schedule.
//
// Stopping: err <= tol * |theta_1| (tol = 1e-12 by default, i.e. 1000x tighter than the
// 1e-9 parity target against ARPACK) and the Ritz value moved by < 1e3 tol |theta_1| since the
// previous check.  No atomics anywhere: results are bit-reproducible and independent of how the
// etas are batched.
//
// History (measured, then removed -- profiles/r03_wide_blocks_ab.json): the
// single-vector recurrence of round 1 (41.1 passes per eta against 31.6, 1079-1094 eta/s against
// 1183-1233) and the four- / eight-vector families of round 2 (vector-FMA quarter strips 587 eta/s,
// matrix-core strips 960, matrix-core bands 1163 with four vectors and 967 with eight, against 1201
// for this kernel in the same interleaved A/B on one MI355X).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <vector>

#include "packed.hpp"
#include "prof.hpp"

namespace scint {

constexpr int kCheckEveryBlock = 2;   // passes per chunk (between convergence checks); SCINT_CHECK_EVERY overrides (tests).
                                      // Measured at 4096^2 / 256 eta with the check on the group's own stream (round 3,
                                      // interleaved, two rounds, profiles/r03_check_stream_ab.txt): every 1 / 2 / 3 passes ->
                                      // 30.6 / 31.2 / 31.6 passes per eta, 1331 / 1412 / 1400 eta/s (check on the sweep's
                                      // stream, every 3: 1387).  Round 2, check in line: every 2 / 3 / 4 / 5 -> 1203 / 1211 / 1196 / 1198.
constexpr int kFirstCheck = 8;
constexpr int kMaxK = 512;   // upper bound on the Lanczos steps a caller may ask for

// Element (row, col) of the packed Hermitian matrix, row != col blocks handled by symmetry.
__device__ inline cplx packed_at(const PackedJob& jb, int r, int c) {
    const int br = r / kTB, bc = c / kTB;
    if (bc >= br) return jb.tiles[(tile_offset(jb.nb, br) + (bc - br)) * kTileElems + (r % kTB) * kTB + (c % kTB)];
    return conj(jb.tiles[(tile_offset(jb.nb, bc) + (br - bc)) * kTileElems + (c % kTB) * kTB + (r % kTB)]);
}

constexpr int kRedGroups = 4;    // wavefronts per reduce block: each sums every 4th partial vector.  8 KiB of LDS: the block fits
                                 // beside the two 76-KiB mat-vec workgroups of a CU.  With 16 wavefronts (32 KiB) it had to wait for
                                 // a mat-vec workgroup of the OTHER slot group to retire: 22 us alone, 170-270 us in the sweep;
                                 // interleaved A/B (profiles/r03_check_stream_ab.txt): 16 / 8 / 4 wavefronts -> 1433 / 1433-1455 / 1454 eta/s


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

// ==============================================================================
// Two-vector (block) Lanczos -- the eigenvalue-only sweep
// ==============================================================================
// Every Lanczos step streams the matrix once; its cost does not depend on how many vectors are
// multiplied on the way (the kernel stays far below the fp64 rate: 2 flop/byte with two
// vectors).  With a block of TWO vectors the Krylov space grows by two dimensions per pass and
// the top Ritz value reaches a given accuracy in ~0.7x the passes of the single-vector
// recurrence on theta-theta matrices (lambda_2 close to lambda_1 stops hurting; measured on the
// CPU at N = 1023: 42/34/27/17/27/29/31 -> 29/27/20/13/20/23/23 passes over the eta range).
//
//   Q_0 R = [v0 v1]            v0 = row n/2 (Eval_calc's start vector), v1 = another row
//   W_j   = A Q_j - Q_{j-1} B_{j-1}^H
//   A_j   = Q_j^H W_j                       (2x2 Hermitian)
//   W_j - Q_j A_j = Q_{j+1} B_j             (B_j 2x2 upper triangular, Cholesky of the Gram matrix)
//
// T_k = blocktridiag(B_{j-1}; A_j; B_j^H) is Hermitian pentadiagonal; its top two eigenvalues
// come from a Sturm count on the banded LDL^H factorisation, the Ritz vector's last block (for
// the residual ||B_{k-1} s_last||) from inverse iteration on the same factorisation.
//
// Q_j = (W_{j-1} - Q_{j-1} A_{j-1}) B_{j-1}^{-1} is built by pk2_coef_kernel from the previous step's
// vectors and the fixed-order partial sums of A_{j-1} = Q^H W and of the Gram matrix W^H W
// (G' = W^H W - A^H A, as beta^2 = |u|^2 - alpha^2 in the single-vector form).  Vectors are
// stored interleaved, [row][2]; scalar partials as 4 doubles per 64-row block.
struct Blk2 {
    double a11, a22; cplx a12;      // A_{j-1}
    double b11, b22; cplx b12;      // B_{j-1}
    double i11, i22;                // 1/b11, 1/b22; 0 when that direction is exhausted
};

// direct: `up` holds the Gram matrix of W - Q A itself (pk2_cert_resid_kernel: step 0 of a certificate), not of W
__device__ inline Blk2 step_block_wave(const double* __restrict__ ap, const double* __restrict__ up, int nb, int lane,
                                       bool direct = false) {
    double s[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = 0.0;
    for (int i = lane; i < nb; i += 64) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[c] += gload(ap + 4 * i + c); s[4 + c] += gload(up + 4 * i + c); }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = wave_sum(s[c]);
    Blk2 b;
    b.a11 = s[0]; b.a22 = s[1]; b.a12 = mk(s[2], s[3]);
    const double n12 = s[2] * s[2] + s[3] * s[3], tr = s[0] + s[1];
    const double h11 = direct ? s[4] : s[4] - (s[0] * s[0] + n12), h22 = direct ? s[5] : s[5] - (n12 + s[1] * s[1]);
    const cplx h12 = direct ? mk(s[6], s[7]) : mk(s[6] - s[2] * tr, s[7] - s[3] * tr);
    // Both pivots are differences of nearly equal numbers once a direction is exhausted (W^H W - A^H A of a
    // vector that lies in the span already built): rounding noise of size 1e-16 |W|^2, which 1 / b would blow
    // up into a garbage basis vector.  A pivot below 1e-14 of its own Gram entry counts as zero.
    // (direct: the entries ARE the squared norms -- no cancellation to guard against, only a true zero)
    const double kPivotFloor = direct ? 0.0 : 1e-14;
    b.b11 = h11 > kPivotFloor * s[4] ? sqrt(h11) : 0.0;
    b.i11 = b.b11 > 0.0 ? 1.0 / b.b11 : 0.0;
    b.b12 = mk(h12.x * b.i11, h12.y * b.i11);
    const double d = h22 - norm2(b.b12);
    b.b22 = d > kPivotFloor * s[5] ? sqrt(d) : 0.0;
    b.i22 = b.b22 > 0.0 ? 1.0 / b.b22 : 0.0;
    return b;
}

// row of Q_j = (W_{j-1} - Q_{j-1} A_{j-1}) B_{j-1}^{-1} from the rows (u1, u2) of W_{j-1}, (q1, q2) of Q_{j-1}
__device__ inline void blk_q_row(const Blk2& b, cplx u1, cplx u2, cplx q1, cplx q2, cplx& x1, cplx& x2) {
    const cplx y1 = u1 - (q1 * b.a11 + mulc(q2, b.a12));       // q1 a11 + q2 conj(a12)
    const cplx y2 = u2 - (q1 * b.a12 + q2 * b.a22);
    x1 = y1 * b.i11;
    x2 = (y2 - x1 * b.b12) * b.i22;
}

// [v0 v1]: rows n/2 and n/2+7 (or a neighbouring row in a tiny matrix) of theta-theta
__device__ inline int second_start_row(int n) { return (n / 2 + 7 < n) ? n / 2 + 7 : (n / 2 + 1) % max(n, 1); }

__global__ void __launch_bounds__(64) pk2_init_kernel(const PackedJob* jobs, const int32_t* slots) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K == 0 && e == 0) { jb.state[1] = 0; jb.result[1] = -INFINITY; jb.result[3] = -INFINITY; }
    if (K >= jb.nb) return;
    const int r = K * kTB + e;
    cplx v0 = mk(0.0, 0.0), v1 = mk(0.0, 0.0);
    if (r < jb.n && jb.n >= 2) { v0 = packed_at(jb, jb.n / 2, r); v1 = packed_at(jb, second_start_row(jb.n), r); }
    jb.U[0][2 * r] = v0; jb.U[0][2 * r + 1] = v1;
    jb.U[1][2 * r] = mk(0.0, 0.0); jb.U[1][2 * r + 1] = mk(0.0, 0.0);
    cplx* qm1 = jb.Q + (int64_t)(jb.qslots - 1) * jb.qstride * 2;     // "Q_{-1}" = 0
    qm1[2 * r] = mk(0.0, 0.0); qm1[2 * r + 1] = mk(0.0, 0.0);
    jb.Q[2 * r] = mk(0.0, 0.0); jb.Q[2 * r + 1] = mk(0.0, 0.0);
    const double g11 = wave_sum(norm2(v0)), g22 = wave_sum(norm2(v1));
    const cplx g12 = wave_sum(mulc(v1, v0));                           // conj(v0) v1
    if (e == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { jb.apart[0][4 * K + c] = 0.0; jb.apart[1][4 * K + c] = 0.0; jb.upart[1][4 * K + c] = 0.0; }
        jb.upart[0][4 * K] = g11; jb.upart[0][4 * K + 1] = g22; jb.upart[0][4 * K + 2] = g12.x; jb.upart[0][4 * K + 3] = g12.y;
    }
}

// The step's 2x2 coefficients, computed ONCE per job and step (pk2_coef_kernel) and read back as ten
// scalars by the consumers (reduce: B_{j-1}).  Layout of jb.coef + 16 * parity:
//   a11 a22 a12.x a12.y  b11 b22 b12.x b12.y  i11 i22
__device__ inline Blk2 load_blk(const double* __restrict__ c) {
    Blk2 b;
    b.a11 = gload(c + 0); b.a22 = gload(c + 1); b.a12 = mk(gload(c + 2), gload(c + 3));
    b.b11 = gload(c + 4); b.b22 = gload(c + 5); b.b12 = mk(gload(c + 6), gload(c + 7));
    b.i11 = gload(c + 8); b.i22 = gload(c + 9);
    return b;
}

// Step j, first launch: A_{j-1}, B_{j-1} from the fixed-order partial sums of the previous reduce
// kernel, then Q_j = (W_{j-1} - Q_{j-1} A_{j-1}) B_{j-1}^{-1} for 1024 rows per workgroup.  Every
// workgroup of a job recomputes the (identical) coefficients; the first one also writes them out.
// Until round 3 every mat-vec workgroup did this itself -- five dependent memory round trips before
// its tile loads could start, 18 % of the streaming rate (5.0 TB/s in the sweep against 6.2 TB/s for
// the same loop on synthetic strips, profiles/r03_pk2_probe.txt).
constexpr int kCoefRows = 1024;
__global__ void __launch_bounds__(kCoefRows) pk2_coef_kernel(const PackedJob* __restrict__ jobs, int launch) {
    __shared__ double sh[16];
    const PackedJob jb = jobs[blockIdx.y];
    const int step = launch - jb.start;
    if (jb.n < 2 || step < 0 || step >= jb.max_steps || gload(jb.state) >= jb.gen) return;
    if ((int)blockIdx.x * kCoefRows >= jb.nb * kTB) return;
    const int par = step & 1;
    if (threadIdx.x < 64) {
        const Blk2 sc = step_block_wave(par ? jb.apart[1] : jb.apart[0], par ? jb.upart[1] : jb.upart[0], jb.nb, threadIdx.x,
                                        jb.certify && step == 1);
        if (threadIdx.x == 0) {
            sh[0] = sc.a11; sh[1] = sc.a22; sh[2] = sc.a12.x; sh[3] = sc.a12.y;
            sh[4] = sc.b11; sh[5] = sc.b22; sh[6] = sc.b12.x; sh[7] = sc.b12.y;
            sh[8] = sc.i11; sh[9] = sc.i22;
            if (blockIdx.x == 0) {
                double* c = jb.coef + 16 * par;
#pragma unroll
                for (int i = 0; i < 10; ++i) gstore(c + i, sh[i]);
                if (step > 0) {
                    double* A = jb.alpha + 4 * (step - 1);
                    A[0] = sc.a11; A[1] = sc.a22; A[2] = sc.a12.x; A[3] = sc.a12.y;
                }
                double* B = jb.beta + 4 * step;                      // B[step] couples blocks step-1 and step
                B[0] = sc.b11; B[1] = sc.b22; B[2] = sc.b12.x; B[3] = sc.b12.y;
            }
        }
    }
    __syncthreads();
    Blk2 sc;
    sc.a11 = sh[0]; sc.a22 = sh[1]; sc.a12 = mk(sh[2], sh[3]);
    sc.b11 = sh[4]; sc.b22 = sh[5]; sc.b12 = mk(sh[6], sh[7]);
    sc.i11 = sh[8]; sc.i22 = sh[9];
    const int r = (int)blockIdx.x * kCoefRows + (int)threadIdx.x;
    if (r >= jb.nb * kTB) return;
    const cplx* __restrict__ Up = par ? jb.U[1] : jb.U[0];
    const int qs = jb.qslots;
    const cplx* __restrict__ Qp = jb.Q + (int64_t)((step + qs - 1) % qs) * jb.qstride * 2;
    cplx* __restrict__ Qn = jb.Q + (int64_t)(step % qs) * jb.qstride * 2;
    cplx x1, x2;
    blk_q_row(sc, gload(Up + 2 * r), gload(Up + 2 * r + 1), gload(Qp + 2 * r), gload(Qp + 2 * r + 1), x1, x2);
    gstore(Qn + 2 * r, x1); gstore(Qn + 2 * r + 1, x2);
}

// One workgroup = one strip: <= 16 column tiles of block row I and (nrows = 2) of block row I+1.  Wave w owns
// the 16-column slice 16w .. 16w+15 of every tile, ALL 64 rows.  A wave load is eight 128-byte row segments:
// lane l = 8 rg + cg reads row 8j + rg, columns 16w + 8cc + cg for the sixteen (j, cc), j = 0..7 row steps,
// cc = 0..1 column chunks.  So
//   * the COLUMN part  c_J[col] = sum_rows conj(a[row][col]) x_I[row]  is lane-local over 8 rows and
//     finishes inside the wave (three shuffle steps over the eight row groups): no barrier.  It goes to LDS
//     -- a wave reads and writes only its own 16-column slice -- where the second row of the pair adds its
//     share, and the strip's partials leave in ONE coalesced burst at the end;
//   * the ROW part  y_I[row] = sum_J sum_col a[row][col] x_J[col]  keeps 8 rows x 2 vectors of per-lane
//     accumulators for a row of the strip and is reduced at its end (8 lanes by shuffles, the four waves
//     through LDS): one barrier per row.
// What round 3 measured on synthetic strips (profiles/r03_pk2_probe.txt, r03_pk2e_probe.txt), each the reason
// for a line above: a 4-wave LDS flush of the column partials with two barriers every 4 tiles (the round-2
// shape: a wave owned 16 rows x 64 columns) took the loop from 6.56 to 5.51 TB/s; a memory operation under a
// branch makes the compiler's wait counts pessimistic for everything behind it, so the loop is branch-free
// (last tile peeled: no conditional prefetch) and carries scheduling fences (left alone, the compiler sinks
// the prefetch below the second half of the tile); and the partial vectors' 3 % of WRITE traffic costs 13 %
// of the rate wherever in the kernel it is issued (6.78 TB/s without it, the same into an L2-resident
// region) -- hence two rows per workgroup: half of it.
// x_I: eight distinct rows per instruction, read from LDS.  256 threads, 76 KiB of LDS, 196 registers: two
// workgroups per CU.
constexpr int kLdsXs = 0, kLdsCol = kMaxStrip * kTB * 2, kLdsXi = 2 * kLdsCol, kLdsRsum = kLdsXi + kRows64 * kTB * 2,
              kLdsElems = kLdsRsum + 4 * kTB * 2;   // complex elements: X_J blocks, column partials, X_I of the rows, row sums of the four waves
constexpr size_t kMatvecLdsBytes = sizeof(cplx) * kLdsElems;
static_assert(kMatvecLdsBytes <= 80 * 1024, "two mat-vec workgroups per CU (160 KiB of LDS)");

__device__ __forceinline__ void pk2_half(const cplx (&a)[8], int h, const cplx (*__restrict__ xir)[2],
                                        const cplx (&xJ1)[2], const cplx (&xJ2)[2], cplx (&acc1)[8], cplx (&acc2)[8],
                                        cplx (&c1)[2], cplx (&c2)[2]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;
        const cplx x1 = xir[8 * j][0], x2 = xir[8 * j][1];             // row 8j + rg of X_I
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const cplx e = a[2 * jj + cc];
            acc1[j] = acc1[j] + e * xJ1[cc];
            acc2[j] = acc2[j] + e * xJ2[cc];
            c1[cc] = mk(c1[cc].x + e.x * x1.x + e.y * x1.y, c1[cc].y + e.x * x1.y - e.y * x1.x);   // conj(a) x_I
            c2[cc] = mk(c2[cc].x + e.x * x2.x + e.y * x2.y, c2[cc].y + e.x * x2.y - e.y * x2.x);
        }
    }
}
// the eight row groups of a column (lanes l ^ 8, ^ 16, ^ 32; fixed order), then this lane's value of the tile's
// [64][2] column partial: row groups 0..3 hold (chunk, vector) = (rg >> 1, rg & 1), groups 4..7 the same again
__device__ __forceinline__ cplx pk2_colsum(cplx (&c1)[2], cplx (&c2)[2], int rg) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            c1[cc] = mk(c1[cc].x + __shfl_xor(c1[cc].x, o, 64), c1[cc].y + __shfl_xor(c1[cc].y, o, 64));
            c2[cc] = mk(c2[cc].x + __shfl_xor(c2[cc].x, o, 64), c2[cc].y + __shfl_xor(c2[cc].y, o, 64));
        }
    }
    const bool v1 = rg & 1, ch1 = rg & 2;        // (component-wise selects: v_cndmask, not an indexed array)
    const double lx = v1 ? c2[0].x : c1[0].x, ly = v1 ? c2[0].y : c1[0].y;
    const double hx = v1 ? c2[1].x : c1[1].x, hy = v1 ? c2[1].y : c1[1].y;
    return mk(ch1 ? hx : lx, ch1 ? hy : ly);
}

// One block row of the strip: tiles t = t0 .. ntile-1 at tp + (t - t0) tiles; a0 holds the first half of tile t0.
// ADD: the column partials are added to what the first row left (the tile tskip -- the second row's diagonal
// tile -- adds nothing); else they are stored.  Then the row partials: 8 lanes by shuffles, 4 waves through LDS.
template <bool ADD>
__device__ __forceinline__ void pk2_row(const cplx* __restrict__ tp, cplx (&a0)[8], int t0, int ntile, int tskip,
                                       const cplx (*__restrict__ xs)[kTB][2], const cplx (*__restrict__ xir)[2],
                                       cplx* __restrict__ cslot, cplx* __restrict__ scratch, cplx (*__restrict__ rsum)[kTB][2],
                                       cplx* __restrict__ rowpart, int lane, int w, int col, int cg, int rg) {
    cplx a1[8], acc1[8], acc2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc1[j] = mk(0.0, 0.0); acc2[j] = mk(0.0, 0.0); }
#pragma unroll 1
    for (int t = t0; t + 1 < ntile; ++t) {
        const cplx* __restrict__ tc = tp + (int64_t)(t - t0) * kTileElems;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = gload_nt(tc + (8 * (4 + (k >> 1))) * kTB + 8 * (k & 1));   // rows 32 .. 63
        cplx xJ1[2], xJ2[2], c1[2], c2[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            xJ1[cc] = xs[t][col + 8 * cc][0]; xJ2[cc] = xs[t][col + 8 * cc][1];
            c1[cc] = mk(0.0, 0.0); c2[cc] = mk(0.0, 0.0);
        }
        pk2_half(a0, 0, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) a0[k] = gload_nt(tc + kTileElems + (8 * (k >> 1)) * kTB + 8 * (k & 1));   // rows 0 .. 31 of the next tile
        __builtin_amdgcn_sched_barrier(0);
        pk2_half(a1, 1, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        cplx c = pk2_colsum(c1, c2, rg);
        if (!ADD) cslot[2 * (t * kTB)] = c;
        else {
            // one read-modify-write per slot: row groups 4..7 (same values as 0..3) go to a scratch element each.
            // An address select, not a branch: `if (rg < 4)` here cost 23 % of the kernel's rate (5.6 -> 4.3 TB/s)
            cplx* __restrict__ dst = rg < 4 ? cslot + 2 * (t * kTB) : scratch;
            const cplx o = *dst;
            const double keep = t == tskip ? 0.0 : 1.0;
            *dst = mk(o.x + keep * c.x, o.y + keep * c.y);
        }
    }
    {   // the last tile: nothing left to prefetch
        const int t = ntile - 1;
        const cplx* __restrict__ tc = tp + (int64_t)(t - t0) * kTileElems;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = gload_nt(tc + (8 * (4 + (k >> 1))) * kTB + 8 * (k & 1));
        cplx xJ1[2], xJ2[2], c1[2], c2[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            xJ1[cc] = xs[t][col + 8 * cc][0]; xJ2[cc] = xs[t][col + 8 * cc][1];
            c1[cc] = mk(0.0, 0.0); c2[cc] = mk(0.0, 0.0);
        }
        pk2_half(a0, 0, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        pk2_half(a1, 1, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        cplx c = pk2_colsum(c1, c2, rg);
        if (!ADD) cslot[2 * (t * kTB)] = c;
        else {
            // one read-modify-write per slot: row groups 4..7 (same values as 0..3) go to a scratch element each.
            // An address select, not a branch: `if (rg < 4)` here cost 23 % of the kernel's rate (5.6 -> 4.3 TB/s)
            cplx* __restrict__ dst = rg < 4 ? cslot + 2 * (t * kTB) : scratch;
            const cplx o = *dst;
            const double keep = t == tskip ? 0.0 : 1.0;
            *dst = mk(o.x + keep * c.x, o.y + keep * c.y);
        }
    }
    // row partials: the 8 lanes of a row group (xor 1, 2, 4; fixed order), then the four waves (column slices)
    if (ADD) __syncthreads();                    // the first row's totals have been read by everybody
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cplx s = v ? acc2[j] : acc1[j];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) s = mk(s.x + __shfl_xor(s.x, o, 64), s.y + __shfl_xor(s.y, o, 64));
            if (cg == 0) rsum[w][8 * j + rg][v] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kTB) {
        const int row = threadIdx.x >> 1, v = threadIdx.x & 1;
        const cplx tot = ((rsum[0][row][v] + rsum[1][row][v]) + rsum[2][row][v]) + rsum[3][row][v];
        gstore(rowpart + threadIdx.x, tot);
    }
}

// (the workgroup's whole work, so that the mixed sweep can put complex128 and complex64 strips into ONE launch:
// pk2_matvec_mixed_kernel in matvec32.hpp)
__device__ __forceinline__ void pk2_matvec_body(const Strip* __restrict__ sp, int launch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];             // kMatvecLdsBytes = 76 KiB: two workgroups per CU
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    cplx (*xs)[kTB][2] = reinterpret_cast<cplx (*)[kTB][2]>(lds + kLdsXs);     // [kMaxStrip]: the blocks X_J = rows of Q_j (32 KiB)
    cplx (*rsum)[kTB][2] = reinterpret_cast<cplx (*)[kTB][2]>(lds + kLdsRsum); // [4 waves][64 rows][2] (8 KiB)
    const int step = launch - sp->start;
    if (step < 0 || step >= sp->max_steps) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3, col = 16 * w + cg;
    const int ntile = sp->ntile;
    const int lane_off = rg * kTB + col;                                        // row rg, first column of the lane
    const cplx* __restrict__ tp = sp->tiles[0] + lane_off;
    cplx a0[8];                                                                 // element 2 jj + cc: row 8 (4h + jj) + rg, column col + 8 cc
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = gload_nt(tp + (8 * (k >> 1)) * kTB + 8 * (k & 1));
    const int32_t done = gload(sp->state);
    const cplx* __restrict__ X = sp->Q + (int64_t)(step % sp->qslots) * sp->qstride * 2;   // Q_j
    const int I = sp->I, J0 = sp->J0, nrows = sp->nrows;
    // X_I of both rows and the strip's X_J blocks: contiguous copies of rows of Q_j (the first tile's loads stay in flight)
    for (int idx = threadIdx.x; idx < nrows * 2 * kTB; idx += 256) lds[kLdsXi + idx] = gload(X + 2 * I * kTB + idx);
    for (int idx = threadIdx.x; idx < ntile * 2 * kTB; idx += 256) lds[kLdsXs + idx] = gload(X + 2 * J0 * kTB + idx);
    if (done >= sp->gen) return;                 // finished job (workgroup-uniform): its loads were harmless
    __syncthreads();
    const cplx (*__restrict__ xi)[2] = reinterpret_cast<const cplx (*)[2]>(lds + kLdsXi);
    // this lane's slot in a tile's [64][2] column partial: column col + 8 (rg >> 1 & 1), vector rg & 1
    cplx* __restrict__ cslot = lds + kLdsCol + 2 * (col + 8 * ((rg >> 1) & 1)) + (rg & 1);
    pk2_row<false>(tp, a0, 0, ntile, -1, xs, xi + rg, cslot, nullptr, rsum, sp->rowpart[0], lane, w, col, cg, rg);
#pragma unroll 1
    for (int r = 1; r < nrows; ++r) {
        // block row I+r over the same columns: its tiles start at column max(J0, I+r); its diagonal tile (I+r, I+r)
        // adds no column partial
        const int t0 = max(0, I + r - J0);
        if (t0 < ntile) {
            const cplx* __restrict__ tpB = sp->tiles[r] + lane_off;
#pragma unroll
            for (int k = 0; k < 8; ++k) a0[k] = gload_nt(tpB + (8 * (k >> 1)) * kTB + 8 * (k & 1));
            // (scratch elements of the upper row groups: the first row's X_I block, dead since that row's barrier)
            pk2_row<true>(tpB, a0, t0, ntile, I + r - J0, xs, xi + r * kTB + rg, cslot, lds + kLdsXi + 32 * w + (lane & 31), rsum,
                          sp->rowpart[r], lane, w, col, cg, rg);
        } else if (threadIdx.x < 2 * kTB) {
            gstore(sp->rowpart[r] + threadIdx.x, mk(0.0, 0.0));   // short strips (tests): row I+r has nothing in this column range
        }
    }
    // the strip's column partials in one burst (the slot of a diagonal tile is written too; nobody reads it)
    __syncthreads();
    cplx* __restrict__ colpart = sp->colpart;
    for (int idx = threadIdx.x; idx < ntile * 2 * kTB; idx += 256) gstore_nt(colpart + idx, lds[kLdsCol + idx]);
}

__global__ void __launch_bounds__(256, 2)
pk2_matvec_kernel(const Strip* __restrict__ strips, int launch) {
    pk2_matvec_body(strips + blockIdx.x, launch);
}

}  // namespace scint
#include "matvec32.hpp"
namespace scint {

__global__ void __launch_bounds__(64 * kRedGroups)
pk2_reduce_kernel(const PackedJob* __restrict__ jobs, int launch) {
    __shared__ cplx part[kRedGroups][kTB][2];
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x;
    const int step = launch - jb.start;
    if (K >= jb.nb || jb.n < 2 || step < 0 || step >= jb.max_steps || gload(jb.state) >= jb.gen) return;
    const int par = step & 1;
    const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
    // fixed summation order: the row partials of block row K strip by strip, then the column partials of
    // the row GROUPS (pairs; quadruples for complex64 strips) above the diagonal in column K; the block's wavefronts
    // take every kRedGroups-th
    const int s0 = jb.row_strip0[K], nrow = jb.row_strip0[K + 1] - s0;
    cplx acc1 = mk(0.0, 0.0), acc2 = mk(0.0, 0.0);
    const int lg = jb.rowgroup_lg;
    const int npair = (K + (1 << lg) - 1) >> lg;                    // groups of block rows (R p .. R p + R-1) with a tile above (K, K)
    for (int idx = g; idx < nrow + npair; idx += kRedGroups) {
        const int cI = (idx - nrow) << lg;                          // column partial of rows cI .. cI+R-1, in tile (cI, K)'s slot
        const cplx* src = idx < nrow ? jb.rowpart + 2 * ((int64_t)(s0 + idx) * kTB + e)
                                     : jb.colpart + 2 * ((tile_offset(jb.nb, cI) + (K - cI)) * kTB + e);
        acc1 = acc1 + gload(src);
        acc2 = acc2 + gload(src + 1);
    }
    part[g][e][0] = acc1;
    part[g][e][1] = acc2;
    __syncthreads();
    if (g == 0) {
        const Blk2 sc = load_blk(jb.coef + 16 * par);               // B_{j-1} (pk2_coef_kernel of this step)
        cplx tot1 = part[0][e][0], tot2 = part[0][e][1];
#pragma unroll
        for (int k = 1; k < kRedGroups; ++k) { tot1 = tot1 + part[k][e][0]; tot2 = tot2 + part[k][e][1]; }
        const int r = K * kTB + e;
        const cplx* __restrict__ Qp = jb.Q + (int64_t)((step + jb.qslots - 1) % jb.qslots) * jb.qstride * 2;
        const cplx* __restrict__ Qn = jb.Q + (int64_t)(step % jb.qslots) * jb.qstride * 2;
        cplx* __restrict__ Un = par ? jb.U[0] : jb.U[1];
        const cplx q1 = gload(Qp + 2 * r), q2 = gload(Qp + 2 * r + 1);
        const cplx x1 = gload(Qn + 2 * r), x2 = gload(Qn + 2 * r + 1);       // row of Q_j
        // row of W_j = A Q_j - Q_{j-1} B_{j-1}^H:  (q1 b11 + q2 conj(b12), q2 b22)
        const cplx t1 = tot1 - (q1 * sc.b11 + mulc(q2, sc.b12));
        const cplx t2 = tot2 - q2 * sc.b22;
        gstore(Un + 2 * r, t1); gstore(Un + 2 * r + 1, t2);
        const double pa11 = wave_sum(x1.x * t1.x + x1.y * t1.y);      // Re(conj(x1) t1)
        const double pa22 = wave_sum(x2.x * t2.x + x2.y * t2.y);
        const cplx pa12 = wave_sum(mulc(t2, x1));                     // conj(x1) t2
        const double pg11 = wave_sum(norm2(t1)), pg22 = wave_sum(norm2(t2));
        const cplx pg12 = wave_sum(mulc(t2, t1));                     // conj(t1) t2
        if (e == 0) {
            double* an = par ? jb.apart[0] : jb.apart[1];
            double* un = par ? jb.upart[0] : jb.upart[1];
            an[4 * K] = pa11; an[4 * K + 1] = pa22; an[4 * K + 2] = pa12.x; an[4 * K + 3] = pa12.y;
            un[4 * K] = pg11; un[4 * K + 1] = pg22; un[4 * K + 2] = pg12.x; un[4 * K + 3] = pg12.y;
        }
    }
}

// Step 0 of a certificate pass starts from vectors that are eigenvectors to ~1e-8: R = W_0 - Q_0 A_0 is 1e-8 of W_0, and the
// Gram form  R^H R = W^H W - A^H A  that every other step uses would lose it entirely (relative 1e-16: below rounding, and
// below the pivot floor, which would then declare the direction exhausted and report a zero residual).  Here R is formed row
// by row -- each element w - (q A) carries a relative error of 1e-8, not 1 -- and the partial sums of ITS Gram matrix replace
// those of W for this one step (step_block_wave(direct)): the certificate's residual || A v - theta v || and the block Q_1 of a
// certificate that continues are computed from the float64 matrix to eight digits.  One wavefront per 64-row block.
__global__ void __launch_bounds__(64) pk2_cert_resid_kernel(const PackedJob* __restrict__ jobs, int launch) {
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x, e = threadIdx.x;
    if (!jb.certify || launch - jb.start != 0 || K >= jb.nb || jb.n < 2 || gload(jb.state) >= jb.gen) return;
    // (step 0: parity 0 -- the reduce kernel wrote W_0 to U[1] and the partial sums to apart[1] / upart[1])
    const Blk2 sc = step_block_wave(jb.apart[1], jb.upart[1], jb.nb, e);
    const int r = K * kTB + e;
    const cplx* __restrict__ Q0 = jb.Q;                                 // slot 0
    const cplx u1 = gload(jb.U[1] + 2 * r), u2 = gload(jb.U[1] + 2 * r + 1);
    const cplx q1 = gload(Q0 + 2 * r), q2 = gload(Q0 + 2 * r + 1);
    const cplx y1 = u1 - (q1 * sc.a11 + mulc(q2, sc.a12));             // as blk_q_row
    const cplx y2 = u2 - (q1 * sc.a12 + q2 * sc.a22);
    const double g11 = wave_sum(norm2(y1)), g22 = wave_sum(norm2(y2));
    const cplx g12 = wave_sum(mulc(y2, y1));                            // conj(y1) y2
    // (other blocks may already have replaced their W-Gram sums when this one reads them above: only A_0 -- apart -- is used here)
    if (e == 0) { double* un = jb.upart[1]; un[4 * K] = g11; un[4 * K + 1] = g22; un[4 * K + 2] = g12.x; un[4 * K + 3] = g12.y; }
}

constexpr int kMaxKB = 128;    // block steps held in LDS by the block check kernel (T up to 256 x 256)
constexpr int kSvecStride = 2 * kMaxKB + 4;   // complex elements between the two exported eigenvectors of T_k (jb.svec)

// eigenvalues of the Hermitian pentadiagonal T (diagonal dg, T[i+1][i] = e1[i], T[i+2][i] = e2[i])
// strictly below x: signs of the pivots of the banded LDL^H factorisation of T - x
__device__ inline int band_count(const double* dg, const cplx* e1, const cplx* e2, int n, double x, double tiny) {
    int cnt = 0;
    double r1 = 0.0, r2 = 0.0;               // 1/d_{i-1}, 1/d_{i-2}
    cplx p = mk(0.0, 0.0), q = mk(0.0, 0.0); // M_{i,i-1}, M_{i,i-2}  (M = L D)
    for (int i = 0; i < n; ++i) {
        double d = ((dg[i] - x) - norm2(p) * r1) - norm2(q) * r2;
        if (fabs(d) < tiny) d = -tiny;
        cnt += d < 0.0;
        const cplx below = i >= 1 ? e2[i - 1] : mk(0.0, 0.0);     // M_{i+1,i-1}
        const cplx pn = e1[i] - mulc(below, p) * r1;               // M_{i+1,i}
        q = below;
        p = pn;
        r2 = r1;
        r1 = 1.0 / d;
    }
    return cnt;
}

__device__ inline double band_multisect(const double* dg, const cplx* e1, const cplx* e2, int n, int target,
                                        double lo, double hi, double tiny, int lane) {
    for (int round = 0; round < 48; ++round) {
        const double wdt = hi - lo;
        if (!(wdt > 0.0)) break;
        const double x = lo + wdt * ((double)(lane + 1) / 65.0);
        const int ok = (x > lo && x < hi) ? (band_count(dg, e1, e2, n, x, tiny) >= target) : 0;
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

__global__ void __launch_bounds__(64) pk2_check_kernel(const PackedJob* jobs, int launches_done) {
    __shared__ double dg[2 * kMaxKB + 2];
    __shared__ cplx e1[2 * kMaxKB + 2], e2[2 * kMaxKB + 2];
    __shared__ double fd[2 * kMaxKB + 2];          // pivots of the factorisation used by the inverse iteration
    __shared__ cplx fm[2 * kMaxKB + 2];            // its M_{i+1,i}
    __shared__ cplx sv[2 * kMaxKB + 2];
    const PackedJob jb = jobs[blockIdx.x];
    if (jb.gen <= 0 || jb.state[0] >= jb.gen) return;      // idle slot / finished job
    const int lane = threadIdx.x;
    const int k_done = launches_done - jb.start;           // block steps this job has completed
    if (jb.n < 2) {
        if (lane == 0) {
            jb.state[2] = 0;
            jb.state[0] = jb.gen;
            jb.status_out[0] = SCINT_E_EMPTY;
            jb.eig_out[0] = nan("");
            if (jb.iters_out) jb.iters_out[0] = 0;
        }
        return;
    }
    if (k_done < kFirstCheck / 2 && k_done < jb.max_steps && !(jb.certify && k_done >= 1)) return;
    const int k = min(k_done, jb.max_steps);
    const int n = 2 * k;
    // A_{k-1}, B_{k-1} are still in the partials of the last reduce kernel
    const Blk2 last = step_block_wave((k & 1) ? jb.apart[1] : jb.apart[0], (k & 1) ? jb.upart[1] : jb.upart[0], jb.nb, lane,
                                      jb.certify && k == 1);
    for (int j = lane; j < k; j += 64) {
        double a11, a22; cplx a12;
        if (j < k - 1) { const double* A = jb.alpha + 4 * j; a11 = A[0]; a22 = A[1]; a12 = mk(A[2], A[3]); }
        else { a11 = last.a11; a22 = last.a22; a12 = last.a12; }
        dg[2 * j] = a11; dg[2 * j + 1] = a22;
        e1[2 * j] = conj(a12);                     // T[2j+1][2j]
        if (j + 1 < k) {
            const double* B = jb.beta + 4 * (j + 1);   // couples blocks j and j+1: T[2j+2.., 2j..] = B
            e2[2 * j] = mk(B[0], 0.0);             // T[2j+2][2j]   = b11
            e1[2 * j + 1] = mk(B[2], B[3]);        // T[2j+2][2j+1] = b12
            e2[2 * j + 1] = mk(B[1], 0.0);         // T[2j+3][2j+1] = b22
        } else {
            e2[2 * j] = mk(0.0, 0.0); e1[2 * j + 1] = mk(0.0, 0.0); e2[2 * j + 1] = mk(0.0, 0.0);
        }
    }
    __syncthreads();
    double lo = INFINITY, hi = -INFINITY, scale = 0.0;
    for (int i = lane; i < n; i += 64) {
        double off = 0.0;
        if (i >= 1) off += sqrt(norm2(e1[i - 1]));
        if (i >= 2) off += sqrt(norm2(e2[i - 2]));
        if (i + 1 < n) off += sqrt(norm2(e1[i]));
        if (i + 2 < n) off += sqrt(norm2(e2[i]));
        lo = fmin(lo, dg[i] - off);
        hi = fmax(hi, dg[i] + off);
        scale = fmax(scale, fabs(dg[i]) + off);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
        scale = fmax(scale, __shfl_xor(scale, o, 64));
    }
    const double bnorm = sqrt(last.b11 * last.b11 + last.b22 * last.b22 + norm2(last.b12));
    const bool finite = isfinite(lo) && isfinite(hi) && isfinite(bnorm);
    double theta = nan(""), theta2 = -INFINITY, resid = nan(""), err = nan("");
    bool have_vectors = false;      // the factorisation ran: T_k's eigenvectors can be exported
    double tiny = 0.0;
    // Ritz vector by inverse iteration on T - sigma (lane 0), sigma just above the Ritz value: factorisation,
    // two solves from the vector of ones; returns |s|^2 of the (max-scaled) vector left in sv
    auto inverse_iteration = [&](double value) {
        const double sigma = value + 8e-16 * fmax(fabs(value), scale * 1e-3);
        double r1 = 0.0, r2 = 0.0;
        cplx p = mk(0.0, 0.0), q = mk(0.0, 0.0);
        for (int i = 0; i < n; ++i) {
            double d = ((dg[i] - sigma) - norm2(p) * r1) - norm2(q) * r2;
            if (fabs(d) < tiny) d = -tiny;
            const cplx below = i >= 1 ? e2[i - 1] : mk(0.0, 0.0);
            const cplx pn = e1[i] - mulc(below, p) * r1;
            fd[i] = d; fm[i] = pn;
            q = below; p = pn; r2 = r1; r1 = 1.0 / d;
        }
        for (int i = 0; i < n; ++i) sv[i] = mk(1.0, 0.0);
        double nrm = 1.0;
        for (int it = 0; it < 2; ++it) {
            // L y = rhs:  L_{i,i-1} = M_{i,i-1}/d_{i-1},  L_{i,i-2} = e2[i-2]/d_{i-2}
            for (int i = 0; i < n; ++i) {
                cplx y = sv[i];
                if (i >= 1) y = y - (fm[i - 1] * sv[i - 1]) * (1.0 / fd[i - 1]);
                if (i >= 2) y = y - (e2[i - 2] * sv[i - 2]) * (1.0 / fd[i - 2]);
                sv[i] = y;
            }
            for (int i = 0; i < n; ++i) sv[i] = sv[i] * (1.0 / fd[i]);
            // L^H s = z
            for (int i = n - 1; i >= 0; --i) {
                cplx z = sv[i];
                if (i + 1 < n) z = z - mulc(sv[i + 1], fm[i]) * (1.0 / fd[i]);        // conj(L_{i+1,i}) s_{i+1}
                if (i + 2 < n) z = z - mulc(sv[i + 2], e2[i]) * (1.0 / fd[i]);        // conj(L_{i+2,i}) s_{i+2}
                sv[i] = z;
            }
            double mx = 0.0;
            for (int i = 0; i < n; ++i) mx = fmax(mx, fmax(fabs(sv[i].x), fabs(sv[i].y)));
            const double sc = mx > 0.0 && isfinite(mx) ? 1.0 / mx : 0.0;
            nrm = 0.0;
            for (int i = 0; i < n; ++i) { sv[i] = sv[i] * sc; nrm += norm2(sv[i]); }
        }
        return nrm;
    };
    if (finite && scale == 0.0 && bnorm == 0.0) {
        theta = 0.0; theta2 = 0.0; resid = 0.0; err = 0.0;     // all-zero theta-theta
    } else if (finite) {
        tiny = scale * 1e-300 + 1e-300;
        lo = lo - 1e-15 * fabs(lo) - 1e-300;
        hi = hi + 1e-15 * fabs(hi) + 1e-300;
        theta = band_multisect(dg, e1, e2, n, n, lo, hi, tiny, lane);
        if (n >= 2) theta2 = band_multisect(dg, e1, e2, n, n - 1, lo, theta, tiny, lane);
        if (lane == 0) {
            // only the last block of the Ritz vector is needed here: resid = || B_{k-1} s_last || / ||s||
            const double nrm = inverse_iteration(theta);
            const cplx sl0 = sv[n - 2], sl1 = sv[n - 1];
            const cplx rr1 = sl0 * last.b11 + last.b12 * sl1;
            const cplx rr2 = sl1 * last.b22;
            resid = nrm > 0.0 ? sqrt((norm2(rr1) + norm2(rr2)) / nrm) : bnorm;
            if (!isfinite(resid)) resid = bnorm;
            if (jb.want_vec || jb.use32) {          // unit-norm eigenvector of T_k for the Ritz vector
                cplx* out = (cplx*)jb.svec;
                const double inv = nrm > 0.0 ? 1.0 / sqrt(nrm) : 0.0;
                for (int i = 0; i < n; ++i) out[i] = sv[i] * inv;
            }
        }
        resid = __shfl(resid, 0, 64);
        const double gap = theta - theta2;
        err = (gap > resid) ? resid * resid / gap : resid;
        have_vectors = true;
    }
    if (lane == 0) {
        const double prev = jb.result[3];
        const double at = fmax(fabs(theta), 1e-300);
        // (a certificate pass starts from converged Ritz vectors: its first value has nothing to settle from)
        const bool settled = (theta - prev) <= 1e3 * jb.tol * at || (jb.certify && k == 1);
        const bool exact = finite && (n >= jb.n || bnorm == 0.0);
        // eigenvector wanted: same gap-aware rule as the single-vector check (pk_check_kernel)
        const double prev2 = jb.result[1], gap2 = theta - theta2;
        const bool gap_ok = gap2 > 0.0 && fabs(theta2 - prev2) <= 0.02 * gap2;
        const bool vec_ok = (resid <= jb.tol * at) || (gap_ok && settled && resid <= jb.vec_gap_factor * jb.tol * gap2);
        // (the iteration phase of a mixed eigenPAIR sweep runs to the eigenVALUE rule: its vector is finished on the complex128 tiles)
        const bool ok = (jb.want_vec && !jb.use32) ? vec_ok : (err <= jb.tol * at && settled);
        const bool conv = finite && (ok || exact);
        const bool stop = conv || !finite || k >= jb.max_steps;
        jb.result[0] = theta; jb.result[1] = theta2; jb.result[2] = resid; jb.result[3] = theta;
        if (stop) {
            // Iteration phase of the mixed sweep, converged: nothing is reported yet -- the two top Ritz vectors go
            // to the certificate pass on the complex128 tiles (the host restarts the slot when it sees state[2]).
            const bool hand_over = jb.use32 && conv && have_vectors && isfinite(theta);
            if (hand_over) {
                cplx* out = (cplx*)jb.svec + kSvecStride;
                if (n >= 2 && theta2 > -INFINITY) {
                    const double nrm = inverse_iteration(theta2);
                    const double inv = nrm > 0.0 && isfinite(nrm) ? 1.0 / sqrt(nrm) : 0.0;
                    for (int i = 0; i < n; ++i) out[i] = sv[i] * inv;
                } else {
                    for (int i = 0; i < n; ++i) out[i] = mk(0.0, 0.0);
                }
            }
            // an iteration phase that ends here (no convergence within the step budget, non-finite input, all-zero
            // matrix) reports the Ritz value of the scaled complex64 copy, scaled back, with its status
            const double unscale = jb.use32 ? 1.0 / gload(jb.scale32) : 1.0;
            jb.state[1] = k;
            jb.state[2] = hand_over ? 1 : 0;
            __threadfence();
            jb.state[0] = jb.gen;
            if (!hand_over) {
                jb.eig_out[0] = (jb.want_vec ? theta : fabs(theta)) * unscale;   // modeler keeps the sign of w
                if (jb.iters_out) jb.iters_out[0] = k + jb.iters_base;
                jb.status_out[0] = (!finite || !isfinite(theta)) ? SCINT_E_NONFINITE : (conv ? SCINT_OK : SCINT_E_NOCONV);
            }
        }
    }
}

// Ritz vector of finished block jobs: y = sum_j (Q_j[:,0] s_{2j} + Q_j[:,1] s_{2j+1}); normalised by
// pk_ritz_scale_kernel (the partial norms go to the first nb entries of upart[0]).
__global__ void __launch_bounds__(64) pk2_ritz_kernel(const PackedJob* jobs, const int32_t* slots,
                                                      const int64_t* eta_index, cplx* vec_out, int64_t vstride) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K >= jb.nb) return;
    const int k = jb.state[1];
    const int r = K * kTB + e;
    const cplx* __restrict__ sv = (const cplx*)jb.svec;
    cplx y = mk(0.0, 0.0);
    for (int j = 0; j < k; ++j) {
        const cplx* __restrict__ q = jb.Q + (int64_t)j * jb.qstride * 2 + 2 * r;
        y = (y + q[0] * sv[2 * j]) + q[1] * sv[2 * j + 1];
    }
    cplx* out = vec_out + eta_index[blockIdx.y] * vstride;
    if (r < jb.n) out[r] = y;
    const double p = wave_sum(r < jb.n ? norm2(y) : 0.0);
    if (e == 0) jb.upart[0][K] = p;   // the job is finished: its partial arrays are free
}

// Start block of a certificate pass (mixed sweep): the two top Ritz vectors of the iteration phase,
//   y_c = sum_j (Q_j[:,0] s_c[2j] + Q_j[:,1] s_c[2j+1]),  c = 1, 2,   s_c = the eigenvectors of T_k the check exported,
// written where pk2_init_kernel writes two rows of theta-theta: into U[0], with the partial sums of their Gram matrix
// (the coefficient kernel of step 0 orthonormalises the block: Cholesky-QR, as for any start block).  The slot's Q
// history is read here for the last time; a thread clears its own rows of the two slots step 0 reads after it has read
// them.  jb is the NEW description of the slot (certify = 1, iters_base = block steps of the iteration phase).
__global__ void __launch_bounds__(64) pk2_restart_kernel(const PackedJob* jobs, const int32_t* slots) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K == 0 && e == 0) { jb.state[1] = 0; jb.result[1] = -INFINITY; jb.result[3] = -INFINITY; }
    if (K >= jb.nb) return;
    const int r = K * kTB + e;
    const int k = jb.iters_base;
    const cplx* __restrict__ s1 = (const cplx*)jb.svec;
    const cplx* __restrict__ s2 = s1 + kSvecStride;
    cplx y1 = mk(0.0, 0.0), y2 = mk(0.0, 0.0);
    for (int j = 0; j < k; ++j) {
        const cplx* __restrict__ q = jb.Q + (int64_t)j * jb.qstride * 2 + 2 * r;
        const cplx q0 = q[0], q1 = q[1];
        y1 = (y1 + q0 * s1[2 * j]) + q1 * s1[2 * j + 1];
        y2 = (y2 + q0 * s2[2 * j]) + q1 * s2[2 * j + 1];
    }
    if (r >= jb.n) { y1 = mk(0.0, 0.0); y2 = mk(0.0, 0.0); }
    jb.U[0][2 * r] = y1; jb.U[0][2 * r + 1] = y2;
    jb.U[1][2 * r] = mk(0.0, 0.0); jb.U[1][2 * r + 1] = mk(0.0, 0.0);
    cplx* qm1 = jb.Q + (int64_t)(jb.qslots - 1) * jb.qstride * 2;     // "Q_{-1}" = 0
    qm1[2 * r] = mk(0.0, 0.0); qm1[2 * r + 1] = mk(0.0, 0.0);
    jb.Q[2 * r] = mk(0.0, 0.0); jb.Q[2 * r + 1] = mk(0.0, 0.0);
    const double g11 = wave_sum(norm2(y1)), g22 = wave_sum(norm2(y2));
    const cplx g12 = wave_sum(mulc(y2, y1));                           // conj(y1) y2
    if (e == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { jb.apart[0][4 * K + c] = 0.0; jb.apart[1][4 * K + c] = 0.0; jb.upart[1][4 * K + c] = 0.0; }
        jb.upart[0][4 * K] = g11; jb.upart[0][4 * K + 1] = g22; jb.upart[0][4 * K + 2] = g12.x; jb.upart[0][4 * K + 3] = g12.y;
    }
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------
struct SlabLayout {
    size_t tiles, tiles32, U0, U1, Q, svec, rowpart, colpart, apart0, apart1, upart0, upart1,
        coef, alpha, beta, result, total;
    int qslots;
};

// strip length of the complex64 strips of an nb-block matrix (a function of nb only, like strip_len_for)
static int strip_len32_for(int nb) { return std::min(strip_len_for(nb), kMaxStrip32); }

// row partial vectors of a matrix (one per block row and strip): complex128 strips (pairs of rows), complex64 strips
static int max_strips(int nb) {
    const int S = strip_len_for(nb);
    int n = 0;
    for (int I = 0; I < nb; ++I) n += row_strip_count(nb, I, S, kRows64);
    return n;
}
static int max_strips32(int nb) {
    const int S = strip_len32_for(nb);
    int n = 0;
    for (int I = 0; I < nb; ++I) n += row_strip_count(nb, I, S, kRows32);
    return n;
}

static SlabLayout slab_layout(int nbmax, int max_steps, bool want_vec, bool mixed) {
    SlabLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    // the strip count is not monotone in nb across the strip-length thresholds: take the max
    int smax = 0;
    for (int nb = 1; nb <= nbmax; ++nb) smax = std::max(smax, std::max(max_strips(nb), mixed ? max_strips32(nb) : 0));
    L.tiles = take(sizeof(cplx) * (size_t)tile_count(nbmax) * kTileElems);
    L.tiles32 = mixed ? take(sizeof(c32) * (size_t)tile_count(nbmax) * kTileElems) : 0;
    // vectors, partial vectors and scalar histories of the two-vector (block) recurrence: 2 columns,
    // 4 scalars per coefficient (all small next to the tiles)
    const size_t bw = 2, sc = bw * bw;
    L.U0 = take(sizeof(cplx) * (size_t)nbmax * kTB * bw);
    L.U1 = take(sizeof(cplx) * (size_t)nbmax * kTB * bw);
    // (block steps: <= kMaxKB + 1 are used)
    // (the mixed sweep keeps every Q_j of the iteration phase: its Ritz vectors start the certificate pass)
    L.qslots = want_vec ? max_steps + 1 : (mixed ? std::min(max_steps, kMaxKB) + 1 : 2);
    L.Q = take(sizeof(cplx) * (size_t)nbmax * kTB * (size_t)L.qslots * bw);
    L.svec = take(sizeof(cplx) * (size_t)std::max<int64_t>(bw * (int64_t)(max_steps + 2), 2 * kSvecStride));   // eigenvector(s) of T_k
    L.rowpart = take(sizeof(cplx) * (size_t)smax * kTB * bw);
    L.colpart = take(sizeof(cplx) * (size_t)tile_count(nbmax) * kTB * bw);
    L.apart0 = take(sizeof(double) * (size_t)nbmax * sc);
    L.apart1 = take(sizeof(double) * (size_t)nbmax * sc);
    L.upart0 = take(sizeof(double) * (size_t)nbmax * sc);
    L.upart1 = take(sizeof(double) * (size_t)nbmax * sc);
    L.coef = take(sizeof(double) * 32);
    L.alpha = take(sizeof(double) * (size_t)(max_steps + 2) * sc);
    L.beta = take(sizeof(double) * (size_t)(max_steps + 3) * sc);
    L.result = take(sizeof(double) * 4);
    L.total = align_up(off, 256);
    return L;
}

constexpr int kTabs = 3;   // rotating copies of the per-chunk tables (job table, strips, slot lists, flags)

struct BatchLayout {
    SlabLayout slab;
    int smax;
    size_t jobs, strips, strips32, states, slots, fin_slots, restart, fin_eta, rs, geoms, scales, scale_bits, total;   // table offsets: copy 0; copies are *_stride apart
    size_t jobs_stride, strips_stride, strips32_stride, list_stride, fin_eta_stride, rs_stride;
};

static BatchLayout batch_layout(int nbmax, int max_steps, int nbatch, bool want_vec, int64_t ncs, bool mixed) {
    BatchLayout B;
    B.slab = slab_layout(nbmax, max_steps, want_vec, mixed);
    B.smax = 0;
    // (workgroups per matrix: never more than its row partials, whichever strip shape is in use)
    for (int nb = 1; nb <= nbmax; ++nb) B.smax = std::max(B.smax, std::max(max_strips(nb), mixed ? max_strips32(nb) : 0));
    size_t off = B.slab.total * (size_t)nbatch;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    B.jobs_stride = align_up(sizeof(PackedJob) * (size_t)nbatch, 256);
    B.strips_stride = align_up(sizeof(Strip) * (size_t)nbatch * (size_t)B.smax, 256);
    B.strips32_stride = mixed ? align_up(sizeof(Strip32) * (size_t)nbatch * (size_t)B.smax, 256) : 0;
    B.list_stride = align_up(sizeof(int32_t) * (size_t)nbatch, 256);
    B.fin_eta_stride = align_up(sizeof(int64_t) * (size_t)nbatch, 256);
    B.jobs = take(B.jobs_stride * kTabs);
    B.strips = take(B.strips_stride * kTabs);
    B.strips32 = take(B.strips32_stride * kTabs);
    B.states = take(sizeof(int32_t) * 4 * (size_t)nbatch);
    B.slots = take(B.list_stride * kTabs);
    B.fin_slots = take(B.list_stride * kTabs);
    B.restart = take(B.list_stride * kTabs);
    B.fin_eta = take(B.fin_eta_stride * kTabs);
    // first strip of every block row, every slot: [nbatch][nbmax + 1] per table copy (ONE upload per changed chunk; until round 5 the
    // table lived in each slot's slab and every fresh slot cost an upload of its own -- 20 of them when 20 curvatures retire together)
    B.rs_stride = align_up(sizeof(int32_t) * (size_t)nbatch * (size_t)(nbmax + 1), 256);
    B.rs = take(B.rs_stride * kTabs);
    B.geoms = take(sizeof(GeomDev) * (size_t)ncs);
    B.scales = take(sizeof(double) * (size_t)ncs);
    B.scale_bits = take(sizeof(unsigned long long) * (size_t)ncs);
    B.total = align_up(off, 256);
    return B;
}


// ---- Mixed precision: "complex64-stored iteration, complex128 certificate" --------------------------------------
// The eigenvalue sweep is bound by the bytes of theta-theta it streams per Lanczos pass.  In this mode the gather also
// writes a complex64 copy of the tiles (scaled by a power of two taken from max |CS|), and a curvature runs in two phases:
//   iteration    block Lanczos exactly as above -- float64 vectors, sums and recurrence -- on the Hermitian matrix
//                A~ = fl32(scale A) (pk2_matvec32_kernel: half the bytes per pass), until ITS top eigenpair has converged
//                by the usual rule (at tol / 2).  A~ is 2^-24-close to scale A entry by entry, so its top eigenvector
//                v~ is within ~1e-8 / gap of A's.  Nothing of this phase is reported.
//   certificate  the two top Ritz vectors [v~, v~_2] start a NEW block-Lanczos run on the complex128 tiles: after ONE
//                pass, T_1 = [v~ v~_2]^H A [v~ v~_2] gives the Ritz value theta (a Rayleigh quotient of the float64
//                matrix: its error is quadratic in the eigenvector error, ~1e-15 relative), theta_2, and the residual
//                || A v - theta v || computed with A itself; the stopping rule of the float64 sweep is applied to THEM
//                (err = resid^2 / (theta - theta_2) <= tol |theta|).  If it holds -- it does unless tol is below what
//                the complex64 perturbation leaves, or the gap is tiny -- theta is returned; if not, the run simply
//                continues on the complex128 tiles until it does.
// So every returned eigenvalue is a Ritz value of the float64 matrix that satisfies the float64 sweep's own a-posteriori
// bound, evaluated in float64 on that matrix; complex64 only chooses the subspace.  Cost per curvature at 4096^2: P passes
// of 68 MB + 1 of 136 MB (+ 68 MB of gather writes) instead of P passes of 136 MB.  Eigenvector sweeps (modeler, chi^2,
// phase retrieval) stay on the float64 tiles throughout: the eigenvector itself is only 1e-8 / gap accurate after the
// iteration phase and would need most of its passes again.
// Selected per process by scint_sweep_precision() (default: SCINT_SWEEP_PRECISION=mixed|f64 in the environment, else f64).
static std::atomic<int>& sweep_mode_ref() {
    static std::atomic<int> mode([] {
        const char* e = getenv("SCINT_SWEEP_PRECISION");
        if (!e || !(e[0] == 'm' || e[0] == 'M')) return 0;
        return strncmp(e + 1, "ixed-all", 8) == 0 ? 2 : 1;     // "mixed-all": the eigenPAIR sweeps too (mode 2 below)
    }());
    return mode;
}
// Mode 2 ("mixed-all", round 4): the eigenPAIR sweeps (modeler, chi^2, retrieval) also iterate on the complex64 copy -- to the
// eigenVALUE rule -- and the run that starts from the two Ritz vectors on the complex128 tiles (the certificate run of the
// eigenvalue sweep) simply continues to the eigenVECTOR rule: the vector returned is a Ritz vector of the float64 matrix that
// meets the float64 sweep's own residual bound.
int sweep_mode() { return sweep_mode_ref().load(); }
static bool mode_is_mixed(int mode, bool want_vec) { return want_vec ? mode == 2 : mode >= 1; }

// (the precision mode is an argument here: run_sweep reads the process-wide switch ONCE and sizes, lays out and runs
//  with that one value -- a concurrent scint_sweep_precision() cannot make the layout disagree with the size check)
static int32_t sweep_workspace_bytes_mode(int64_t M, int64_t neta, int64_t batch, int32_t max_iter, bool want_vec,
                                          int64_t ncs, int mode, size_t* bytes) {
    SCINT_REQUIRE(bytes && M >= 1 && neta >= 1 && batch >= 1 && max_iter >= 1 && ncs >= 1,
                  "sweep_workspace_bytes: bad arguments");
    const int nbmax = (int)ceil_div(M, kTB);
    const int steps = (int)std::min<int64_t>(std::min<int64_t>(max_iter, M), kMaxK);
    const int nbatch = (int)std::min(batch, neta);
    *bytes = batch_layout(nbmax, steps, nbatch, want_vec, ncs, mode_is_mixed(mode, want_vec)).total + 4096;
    return SCINT_OK;
}
int32_t sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch, int32_t max_iter, bool want_vec,
                              int64_t ncs, size_t* bytes) {
    return sweep_workspace_bytes_mode(M, neta, batch, max_iter, want_vec, ncs, sweep_mode(), bytes);
}

// Scheduling switches (scint_sweep_schedule; initial values from SCINT_SWEEP_DEPTH / SCINT_CHECK_EVERY / SCINT_SWEEP_GROUPS,
// read ONCE per process -- until round 4 run_sweep called getenv three times per sweep).  0 = the measured default.  None
// selects a different kernel; depth and groups change no result bit (the check cadence moves the pass a curvature stops at,
// i.e. its value inside the tolerance): they exist for the tests that prove exactly that, and for bench.py's one-slot-group leg.
struct SweepSchedule { std::atomic<int> depth{0}, check_every{0}, groups{0}; };
static SweepSchedule& sweep_schedule() {
    static SweepSchedule sc;
    static const bool init = [] {
        auto env_int = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
        sc.depth = env_int("SCINT_SWEEP_DEPTH"); sc.check_every = env_int("SCINT_CHECK_EVERY"); sc.groups = env_int("SCINT_SWEEP_GROUPS");
        return true;
    }();
    (void)init;
    return sc;
}

// Pinned host staging, kept per host thread and grown on demand (a hipHostMalloc per sweep call
// costs more than a small sweep).  Every table that travels to or from the device while kernels
// are in flight lives here, in kTabs rotating copies.
static char* pinned_staging(size_t bytes) {
    thread_local char* buf = nullptr;
    thread_local size_t cap = 0;
    if (bytes > cap) {
        if (buf) (void)hipHostFree(buf);
        buf = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        if (hipHostMalloc(&buf, want) != hipSuccess) {
            set_error("scint: hipHostMalloc of the sweep staging buffer failed");
            buf = nullptr;
            return nullptr;
        }
        cap = want;
    }
    return buf;
}

// Internal streams, one set per (host thread, device): `aux` drives the second group of slots;
// `tail[kTailLanes]` (chi^2 sweep only) run the per-curvature model steps of the retired curvatures, round robin.
// (Measured in round 3, profiles/r03_tail_schedule_ab.json: highest stream priority for the tail streams costs
// 6 %, 1 / 2 / 4 lanes and 36-KiB / 144-KiB back-map workgroups are within 3 % of each other -- the chi^2 sweep
// is bound by the SUM of the mat-vec's and the tail kernels' GPU time, not by how they interleave.  Round 4,
// profiles/r04_tail_schedule_ab.txt, confirmed it from the other side: LOWEST priority for the tail streams, a
// back-map capped at 256 / 512 / 1024 resident workgroups, 320- and 512-row slabs that fit beside two mat-vec
// workgroups, three or four mat-vec workgroups per CU: all within +-2 % or worse.  What moves the sweep is less
// GPU time in the tail kernels themselves.)
// The sweep's events live here too, created once with the streams (until round 4 run_sweep created and destroyed ~20 of them
// per call): a sweep drains every stream before it returns, so the next sweep of the thread finds them all idle.
constexpr int kTabsEv = 3;   // = kTabs (declared below)
struct SideStreams {
    hipStream_t aux = nullptr, chk[2] = {}, tail[kTailLanes] = {};
    hipEvent_t chunk_done[2][kTabsEv] = {}, export_done[2][kTabsEv] = {}, steps_done[2][kTabsEv] = {};
    hipEvent_t start_ev = nullptr, stagger_ev = nullptr;
};
static SideStreams* side_streams() {
    thread_local std::map<int, SideStreams> streams;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto it = streams.find(dev);
    if (it != streams.end()) return &it->second;
    SideStreams s;
    if (hipStreamCreateWithFlags(&s.aux, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (auto& c : s.chk)
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) return nullptr;
    // (round 5, call 5: tail streams confined to every 2nd / 3rd / 4th compute unit by hipExtStreamCreateWithCUMask lose 3.5 % or more --
    //  profiles/r05_call5_cu_mask.txt: the confined tail falls behind and the mat-vec streams no faster beside it; the code is gone)
    for (auto& t : s.tail)
        if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) return nullptr;
    auto mk_ev = [](hipEvent_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess; };
    for (int g = 0; g < 2; ++g)
        for (int t = 0; t < kTabsEv; ++t)
            if (!mk_ev(&s.chunk_done[g][t]) || !mk_ev(&s.export_done[g][t]) || !mk_ev(&s.steps_done[g][t])) return nullptr;
    if (!mk_ev(&s.start_ev) || !mk_ev(&s.stagger_ev)) return nullptr;
    return &(streams[dev] = s);
}

// ----------------------------------------------------------------------------------------------
// The scheduler.  The `batch` slots are split into two GROUPS, each driven on its own stream (the
// caller's and `aux`): the gaps of one stream -- launch boundaries, the latency-bound reduce and
// check kernels, refills -- are filled by the other stream's kernels.  Inside a group the Lanczos
// steps are queued in CHUNKS of kCheckEveryBlock passes + one convergence check + the read-back of the
// per-slot state words, and the host runs `depth` chunks ahead of the state it has seen (depth 2
// by default): while chunk c executes, chunk c+1 is already queued, so a stream never waits for
// the host.  Retiring and refilling therefore lag: a curvature that converges in chunk k is seen
// when chunk k+2 is prepared, its slot idles for one chunk -- the kernels of chunk k+1 skip it,
// because every job carries the GENERATION of its slot and the state word holds the last finished
// generation -- and the next curvature starts in chunk k+2: table upload, gather and start vector
// are queued on the group's stream in front of that chunk's steps.  Host staging never changes
// under a queued copy: job table, strip list, slot lists and flags exist in kTabs rotating
// copies, on the host and on the device.  Per-job arithmetic does not depend on any of this
// (fixed-order sums inside a job), so results are bit-identical for every depth, batch size,
// grouping and arrival order.
// What the calling thread's last sweep streamed (scint_sweep_stats): algorithmic bytes 4 n (n + 1) per complex64 pass and
// 8 n (n + 1) per complex128 pass of an n x n matrix, certificates started, complex128 passes they took.
static_assert(kTabsEv == kTabs, "event sets per group = table copies");
struct SweepStats { double bytes32 = 0, bytes64 = 0, certified = 0, cert_passes = 0; };
static SweepStats& sweep_stats() { thread_local SweepStats st; return st; }

struct SweepProblem {
    const cplx* cs; int64_t cs_stride; const int32_t* cs_index; const double* th_cents; int64_t M;
    const int32_t* keep_idx; const int32_t* keep_n; const double* etas; int64_t neta;
    double* eigs_out; int32_t* status_out; int32_t* iters_out;
    bool want_vec; cplx* vec_out; int64_t vstride;
    SweepTail* tail_hook; hipStream_t tail[kTailLanes]; int tail_rr = 0;   // retired curvatures go round the tail streams
    int nbmax, steps_cap, depth, check_every;
    bool mixed = false; double tol = 0.0; const double* scales_dev = nullptr;   // the mixed sweep (see "Mixed precision" above)
    char* base; BatchLayout BL; const GeomDev* geoms_dev; int32_t* states_dev;
    int64_t next_eta = 0;                 // the queue of curvatures still to be started (both groups pull)
    hipEvent_t stagger_ev = nullptr;      // recorded after the first group's first pass
};

struct SweepGroup {
    SweepProblem* P;
    int slot0, nslots;                    // global index of the group's first slot, its slot count
    hipStream_t stream;
    hipStream_t check_stream;             // the group's convergence checks run here, beside the next chunk's first pass
    hipEvent_t chunk_done[kTabs], export_done[kTabs], steps_done[kTabs];
    // host staging (pinned), one set per table copy; slot indices in all tables are group-local
    PackedJob* h_jobs[kTabs]; Strip* h_strips[kTabs]; int32_t* h_fresh[kTabs]; int32_t* h_fin[kTabs];
    int64_t* h_fin_eta[kTabs]; int32_t* h_rs[kTabs]; int32_t* h_flags[kTabs];
    Strip32* h_strips32[kTabs]; int32_t* h_restart[kTabs];
    // schedule state
    std::vector<PackedJob> jobs;          // current description of every slot of the group
    std::vector<int64_t> slot_eta;        // running eta or -1
    std::vector<int32_t> slot_gen;
    std::vector<int32_t> fin_slots;
    std::vector<int64_t> fin_eta;
    std::vector<int8_t> slot_phase;       // mixed sweep: 0 = iteration on the complex64 tiles, 1 = certificate on the complex128 tiles
    std::vector<std::pair<int32_t, int32_t>> restart;   // (slot, block steps of its iteration phase): certificate passes to start
    int nstrips32 = 0;
    int active = 0, chunk = 0, seen = 0;  // chunks queued / chunks whose flags have been harvested
    bool finished = false;
    int tab_of_chunk[kTabs];              // table copy used by chunk c, indexed c % kTabs
    int nstrips = 0, nb_run = 1;

    PackedJob* d_jobs(int t) const { return (PackedJob*)(P->base + P->BL.jobs + P->BL.jobs_stride * (size_t)t) + slot0; }
    Strip* d_strips(int t) const {
        return (Strip*)(P->base + P->BL.strips + P->BL.strips_stride * (size_t)t) + (size_t)slot0 * (size_t)P->BL.smax;
    }
    int32_t* d_fresh(int t) const { return (int32_t*)(P->base + P->BL.slots + P->BL.list_stride * (size_t)t) + slot0; }
    int32_t* d_fin(int t) const { return (int32_t*)(P->base + P->BL.fin_slots + P->BL.list_stride * (size_t)t) + slot0; }
    int32_t* d_restart(int t) const { return (int32_t*)(P->base + P->BL.restart + P->BL.list_stride * (size_t)t) + slot0; }
    Strip32* d_strips32(int t) const {
        return (Strip32*)(P->base + P->BL.strips32 + P->BL.strips32_stride * (size_t)t) + (size_t)slot0 * (size_t)P->BL.smax;
    }
    int32_t* d_rs(int t) const {
        return (int32_t*)(P->base + P->BL.rs + P->BL.rs_stride * (size_t)t) + (size_t)slot0 * (size_t)(P->nbmax + 1);
    }
    int64_t* d_fin_eta(int t) const { return (int64_t*)(P->base + P->BL.fin_eta + P->BL.fin_eta_stride * (size_t)t) + slot0; }

    // retire what chunk `c` finished (its flags are on the host)
    void harvest(int c) {
        const int32_t* flags = h_flags[c % kTabs];
        fin_slots.clear();
        fin_eta.clear();
        restart.clear();
        for (int s = 0; s < nslots; ++s) {
            if (slot_eta[(size_t)s] < 0 || flags[4 * s] < slot_gen[(size_t)s]) continue;
            const double nn = (double)jobs[(size_t)s].n * ((double)jobs[(size_t)s].n + 1.0), steps = (double)flags[4 * s + 1];
            SweepStats& st = sweep_stats();
            if (jobs[(size_t)s].use32) st.bytes32 += 4.0 * nn * steps; else st.bytes64 += 8.0 * nn * steps;
            if (slot_phase[(size_t)s] == 1) st.cert_passes += steps;
            if (P->mixed && slot_phase[(size_t)s] == 0 && flags[4 * s + 2] == 1) {
                // the iteration phase has converged: the curvature stays in its slot for the certificate pass
                st.certified += 1;
                slot_phase[(size_t)s] = 1;
                restart.push_back({s, flags[4 * s + 1]});
                continue;
            }
            slot_phase[(size_t)s] = 0;
            fin_slots.push_back(s);
            fin_eta.push_back(slot_eta[(size_t)s]);
            slot_eta[(size_t)s] = -1;            // results were written by the check kernel
            --active;
        }
    }

    // prepare and enqueue chunk `chunk`: refill idle slots, then kCheckEveryBlock passes + check + read-back
    int32_t enqueue(int fin_chunk) {
        SweepProblem& S = *P;
        const SlabLayout& L = S.BL.slab;
        std::vector<int32_t> fresh;
        const int launch0 = chunk * S.check_every;
        for (int s = 0; s < nslots && S.next_eta < S.neta; ++s) {
            if (slot_eta[(size_t)s] >= 0) continue;
            const int64_t e = S.next_eta++;
            slot_eta[(size_t)s] = e;
            ++active;
            PackedJob& J = jobs[(size_t)s];
            const int n = S.keep_n[e];
            J.eta = S.etas[e]; J.two_eta = 2 * S.etas[e];
            const int64_t c = S.cs_index ? S.cs_index[e] : 0;
            J.cs = S.cs + c * S.cs_stride; J.th = S.th_cents + c * S.M; J.geom = (int32_t)c;
            J.keep = S.keep_idx + e * S.M; J.n = n; J.nb = (int)ceil_div(std::max(n, 1), kTB);
            J.max_steps = std::min(std::min(S.steps_cap, kMaxKB), std::max((n + 1) / 2, 1));
            J.strip_len = S.mixed ? strip_len32_for(J.nb) : strip_len_for(J.nb);
            J.start = launch0;
            J.gen = ++slot_gen[(size_t)s];
            J.eig_out = S.eigs_out + e; J.status_out = S.status_out + e;
            J.iters_out = S.iters_out ? S.iters_out + e : nullptr;
            J.use32 = S.mixed ? 1 : 0; J.certify = 0; J.iters_base = 0; J.rowgroup_lg = S.mixed ? kRows32Lg : kRows64Lg;
            J.scale32 = S.mixed ? S.scales_dev + c : nullptr;
            J.tol = S.mixed ? 0.5 * S.tol : S.tol;      // (the certificate then passes at the first attempt: measured 1.00 passes per curvature)
            slot_phase[(size_t)s] = 0;
            fresh.push_back(s);
        }
        // certificate passes: the same curvature in the same slot, a new generation on the complex128 tiles.  It joins at
        // the LAST pass of this chunk, so that the chunk's check sees its first (and, as a rule, only) step.
        for (const auto& rs : restart) {
            PackedJob& J = jobs[(size_t)rs.first];
            J.use32 = 0; J.certify = 1; J.iters_base = rs.second; J.rowgroup_lg = kRows64Lg;
            J.strip_len = strip_len_for(J.nb);
            J.start = launch0 + S.check_every - 1;
            J.gen = ++slot_gen[(size_t)rs.first];
            J.tol = S.tol;
        }
        const bool changed = chunk == 0 || !fresh.empty() || !fin_slots.empty() || !restart.empty();
        int tab = chunk == 0 ? 0 : tab_of_chunk[(chunk - 1) % kTabs];
        hipError_t he = hipSuccess;
        if (changed) {
            // a table copy no queued kernel can still read: not the previous chunk's, and not the one
            // the eigenvector export of the retired jobs reads
            if (chunk > 0) {
                const int busy1 = tab_of_chunk[(chunk - 1) % kTabs];
                const int busy2 = fin_chunk >= 0 ? tab_of_chunk[fin_chunk % kTabs] : busy1;
                for (tab = 0; tab == busy1 || tab == busy2; ++tab) {}
            }
            // strips of every running job, block row by block row; longest strips first: the short
            // ones of the last block rows then fill the tail of the launch (dispatch order only)
            Strip* hs = h_strips[tab];
            Strip32* hs32 = h_strips32[tab];
            int32_t* hrs = h_rs[tab];
            nstrips = 0;
            nstrips32 = 0;
            nb_run = 1;
            for (int s = 0; s < nslots; ++s) {
                PackedJob& J = jobs[(size_t)s];
                if (slot_eta[(size_t)s] < 0) { J.n = 0; continue; }      // idle: nothing to launch over
                nb_run = std::max(nb_run, J.nb);
                int32_t* rs0 = hrs + (size_t)s * (size_t)(S.nbmax + 1);
                int idx = 0;
                const int R = 1 << J.rowgroup_lg;
                for (int I = 0; I < J.nb; ++I) { rs0[I] = idx; idx += row_strip_count(J.nb, I, J.strip_len, R); }
                rs0[J.nb] = idx;
                if (J.n < 2) continue;                             // nothing to multiply (the check kernel reports it)
                if (J.use32) {                                     // rows I .. I+3 together, cut on row I's column grid
                    for (int I = 0; I < J.nb; I += kRows32) {
                        const int nrows = std::min(kRows32, J.nb - I);
                        int k = 0;
                        for (int J0 = I; J0 < J.nb; J0 += J.strip_len, ++k) {
                            Strip32& st = hs32[nstrips32++];
                            st.Q = J.Q; st.qstride = J.qstride; st.qslots = J.qslots;
                            st.colpart = J.colpart + 2 * (tile_offset(J.nb, I) + (J0 - I)) * kTB;
                            st.state = J.state;
                            st.I = I; st.J0 = J0; st.ntile = std::min(J.nb, J0 + J.strip_len) - J0;
                            st.start = J.start; st.gen = J.gen; st.max_steps = J.max_steps; st.nrows = nrows;
                            for (int r = 0; r < kRows32; ++r) {
                                const int Ir = std::min(I + r, J.nb - 1), JB = std::max(J0, Ir);
                                st.tiles[r] = J.tiles32 + (tile_offset(J.nb, Ir) + (JB - Ir)) * kTileElems;
                                st.rowpart[r] = J.rowpart + 2 * (int64_t)(rs0[Ir] + k) * kTB;
                            }
                        }
                    }
                    continue;
                }
                for (int I = 0; I < J.nb; I += kRows64) {          // rows I .. I+kRows64-1 together, cut on row I's column grid
                    const int nrows = std::min(kRows64, J.nb - I);
                    int k = 0;
                    for (int J0 = I; J0 < J.nb; J0 += J.strip_len, ++k) {
                        Strip& st = hs[nstrips++];
                        st.Q = J.Q; st.qstride = J.qstride; st.qslots = J.qslots;
                        st.colpart = J.colpart + 2 * (tile_offset(J.nb, I) + (J0 - I)) * kTB;
                        st.state = J.state;
                        st.I = I; st.J0 = J0; st.ntile = std::min(J.nb, J0 + J.strip_len) - J0;
                        st.start = J.start; st.gen = J.gen; st.max_steps = J.max_steps; st.nrows = nrows;
                        for (int r = 0; r < kRows64; ++r) {
                            const int Ir = std::min(I + r, J.nb - 1), JB = std::max(J0, Ir);
                            st.tiles[r] = J.tiles + (tile_offset(J.nb, Ir) + (JB - Ir)) * kTileElems;
                            st.rowpart[r] = J.rowpart + 2 * (int64_t)(rs0[Ir] + k) * kTB;
                        }
                    }
                }
            }
            std::stable_sort(hs, hs + nstrips, [](const Strip& a, const Strip& b) {
                return a.ntile * a.nrows > b.ntile * b.nrows;
            });
            std::stable_sort(hs32, hs32 + nstrips32, [](const Strip32& a, const Strip32& b) {
                return a.ntile * a.nrows > b.ntile * b.nrows;
            });
            for (const auto& rs : restart) h_restart[tab][&rs - restart.data()] = rs.first;
            // every running job's row-strip table travels with this table copy (one upload; the jobs of the copy point into it)
            for (int s = 0; s < nslots; ++s) jobs[(size_t)s].row_strip0 = d_rs(tab) + (size_t)s * (size_t)(S.nbmax + 1);
            std::copy(jobs.begin(), jobs.end(), h_jobs[tab]);
            std::copy(fresh.begin(), fresh.end(), h_fresh[tab]);
            int nb_fresh = 0;
            for (int s : fresh) nb_fresh = std::max(nb_fresh, jobs[(size_t)s].nb);
            he = hipMemcpyAsync(d_rs(tab), hrs, sizeof(int32_t) * (size_t)nslots * (size_t)(S.nbmax + 1), hipMemcpyHostToDevice, stream);
            if (he == hipSuccess)
                he = hipMemcpyAsync(d_jobs(tab), h_jobs[tab], sizeof(PackedJob) * (size_t)nslots, hipMemcpyHostToDevice, stream);
            if (he == hipSuccess && nstrips > 0)
                he = hipMemcpyAsync(d_strips(tab), hs, sizeof(Strip) * (size_t)nstrips, hipMemcpyHostToDevice, stream);
            if (he == hipSuccess && nstrips32 > 0)
                he = hipMemcpyAsync(d_strips32(tab), hs32, sizeof(Strip32) * (size_t)nstrips32, hipMemcpyHostToDevice, stream);
            if (he == hipSuccess && !restart.empty())
                he = hipMemcpyAsync(d_restart(tab), h_restart[tab], sizeof(int32_t) * restart.size(), hipMemcpyHostToDevice, stream);
            if (he == hipSuccess && !fresh.empty())
                he = hipMemcpyAsync(d_fresh(tab), h_fresh[tab], sizeof(int32_t) * fresh.size(), hipMemcpyHostToDevice, stream);
            if (he != hipSuccess) return hip_fail(he, "sweep table upload", __FILE__, __LINE__);
            // eigenvectors of the retired jobs leave their slots BEFORE the new jobs' start vectors
            // overwrite them (same stream); the table copy of the chunk that finished them still
            // describes them
            if (S.want_vec && !fin_slots.empty()) {
                const int ft = tab_of_chunk[fin_chunk % kTabs];
                int nfin = 0, nb_fin = 1;
                for (size_t k = 0; k < fin_slots.size(); ++k) {
                    const PackedJob& F = h_jobs[ft][fin_slots[k]];
                    if (F.n < 2) continue;
                    h_fin[tab][nfin] = fin_slots[k];
                    h_fin_eta[tab][nfin] = fin_eta[k];
                    nb_fin = std::max(nb_fin, F.nb);
                    ++nfin;
                }
                if (nfin > 0) {
                    he = hipMemcpyAsync(d_fin(tab), h_fin[tab], sizeof(int32_t) * (size_t)nfin, hipMemcpyHostToDevice, stream);
                    if (he == hipSuccess)
                        he = hipMemcpyAsync(d_fin_eta(tab), h_fin_eta[tab], sizeof(int64_t) * (size_t)nfin,
                                            hipMemcpyHostToDevice, stream);
                    if (he != hipSuccess) return hip_fail(he, "sweep eigenvector export", __FILE__, __LINE__);
                    const dim3 grid((unsigned)nb_fin, (unsigned)nfin);
                    hipLaunchKernelGGL(pk2_ritz_kernel, grid, dim3(64), 0, stream, d_jobs(ft), d_fin(tab), d_fin_eta(tab),
                                       S.vec_out, S.vstride);
                    hipLaunchKernelGGL(pk_ritz_scale_kernel, grid, dim3(64), 0, stream, d_jobs(ft), d_fin(tab),
                                       d_fin_eta(tab), S.vec_out, S.vstride);
                }
            }
            if (S.tail_hook && !fin_eta.empty()) {
                he = hipEventRecord(export_done[tab], stream);
                for (int l = 0; l < kTailLanes && he == hipSuccess; ++l) he = hipStreamWaitEvent(S.tail[l], export_done[tab], 0);
                if (he != hipSuccess) return hip_fail(he, "sweep tail hand-off", __FILE__, __LINE__);
                const int bm = std::max(1, S.tail_hook->batch_max());
                for (size_t k0 = 0; k0 < fin_eta.size(); k0 += (size_t)bm) {
                    const int l = S.tail_rr++ % kTailLanes;
                    const int32_t rc = S.tail_hook->retire_batch(fin_eta.data() + k0, (int)std::min<size_t>((size_t)bm, fin_eta.size() - k0),
                                                                 S.tail[l], l);
                    if (rc != SCINT_OK) return rc;
                }
            }
            if (!restart.empty()) {
                int nb_re = 1;
                for (const auto& rs : restart) nb_re = std::max(nb_re, jobs[(size_t)rs.first].nb);
                hipLaunchKernelGGL(pk2_restart_kernel, dim3((unsigned)nb_re, (unsigned)restart.size()), dim3(64), 0, stream,
                                   d_jobs(tab), d_restart(tab));
            }
            if (!fresh.empty()) {
                int32_t rc = launch_gather_packed(S.geoms_dev, S.M, d_jobs(tab), d_fresh(tab), (int)fresh.size(), nb_fresh, stream,
                                                  S.mixed);
                if (rc != SCINT_OK) return rc;
                hipLaunchKernelGGL(pk2_init_kernel, dim3((unsigned)nb_fresh, (unsigned)fresh.size()), dim3(64), 0, stream,
                                   d_jobs(tab), d_fresh(tab));
            }
            he = hipGetLastError();
            if (he != hipSuccess) return hip_fail(he, "sweep refill", __FILE__, __LINE__);
        }
        tab_of_chunk[chunk % kTabs] = tab;
        if (nstrips + nstrips32 > 0) {
            for (int i = 0; i < S.check_every; ++i) {
                const int launch = launch0 + i;
                // the previous chunk's check (on check_stream) reads the partial sums the SECOND reduce from here
                // overwrites: it has had a whole pass to finish, this wait only makes that a guarantee
                if (chunk > 0 && i == (S.check_every > 1 ? 1 : 0))
                    (void)hipStreamWaitEvent(stream, chunk_done[(chunk - 1) % kTabs], 0);
                hipLaunchKernelGGL(pk2_coef_kernel, dim3((unsigned)ceil_div(nb_run * kTB, kCoefRows), (unsigned)nslots),
                                   dim3(kCoefRows), 0, stream, d_jobs(tab), launch);
                if (nstrips32 > 0 && nstrips > 0) {
                    // certificate passes ride in the launch of the complex64 strips (a launch of their own -- a few
                    // curvatures' strips -- ran at 1.4 TB/s and took 17 % of the mixed sweep's mat-vec time)
                    const int slot = profiler().begin(kProfMatvec32, stream);
                    hipLaunchKernelGGL(pk2_matvec_mixed_kernel, dim3((unsigned)(nstrips32 + nstrips)), dim3(256), kMatvecMixedLdsBytes, stream,
                                       d_strips32(tab), nstrips32, d_strips(tab), launch);
                    profiler().end(kProfMatvec32, slot, stream);
                } else if (nstrips32 > 0) {
                    const int slot = profiler().begin(kProfMatvec32, stream);
                    hipLaunchKernelGGL(pk2_matvec32_kernel, dim3((unsigned)nstrips32), dim3(256), kMatvec32LdsBytes, stream,
                                       d_strips32(tab), launch);
                    profiler().end(kProfMatvec32, slot, stream);
                } else if (nstrips > 0) {
                    const int slot = profiler().begin(kProfMatvec, stream);
                    hipLaunchKernelGGL(pk2_matvec_kernel, dim3((unsigned)nstrips), dim3(256), kMatvecLdsBytes, stream, d_strips(tab), launch);
                    profiler().end(kProfMatvec, slot, stream);
                }
                hipLaunchKernelGGL(pk2_reduce_kernel, dim3((unsigned)nb_run, (unsigned)nslots), dim3(64 * kRedGroups), 0,
                                   stream, d_jobs(tab), launch);
                // the other group starts one pass behind this one (see run_sweep): its checks, refills and
                // reductions then fall beside this group's mat-vecs instead of beside its own twins
                if (!restart.empty() && i == S.check_every - 1)      // step 0 of the certificates that joined at this pass
                    hipLaunchKernelGGL(pk2_cert_resid_kernel, dim3((unsigned)nb_run, (unsigned)nslots), dim3(64), 0, stream,
                                       d_jobs(tab), launch);
                if (chunk == 0 && i == 0 && S.stagger_ev && slot0 == 0) (void)hipEventRecord(S.stagger_ev, stream);
            }
        }
        he = hipEventRecord(steps_done[chunk % kTabs], stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(check_stream, steps_done[chunk % kTabs], 0);
        if (he != hipSuccess) return hip_fail(he, "sweep check hand-off", __FILE__, __LINE__);
        hipLaunchKernelGGL(pk2_check_kernel, dim3((unsigned)nslots), dim3(64), 0, check_stream, d_jobs(tab), launch0 + S.check_every);
        he = hipGetLastError();
        if (he == hipSuccess)
            he = hipMemcpyAsync(h_flags[chunk % kTabs], S.states_dev + 4 * slot0, sizeof(int32_t) * 4 * (size_t)nslots,
                                hipMemcpyDeviceToHost, check_stream);
        if (he == hipSuccess) he = hipEventRecord(chunk_done[chunk % kTabs], check_stream);
        if (he != hipSuccess) return hip_fail(he, "sweep chunk", __FILE__, __LINE__);
        ++chunk;
        return SCINT_OK;
    }

    // one scheduling step of the group: wait for its oldest chunk if the pipeline is full (or only
    // draining is left), retire, and queue the next chunk.  Sets `finished` when nothing is left.
    int32_t advance() {
        SweepProblem& S = *P;
        int fin_chunk = -1;
        fin_slots.clear();
        fin_eta.clear();
        restart.clear();
        const bool more = active > 0 || S.next_eta < S.neta;
        if (chunk - seen >= S.depth || (!more && seen < chunk)) {
            hipError_t he = hipEventSynchronize(chunk_done[seen % kTabs]);
            if (he != hipSuccess) return hip_fail(he, "sweep wait", __FILE__, __LINE__);
            if (profiler().enabled) profiler().collect();
            harvest(seen);
            fin_chunk = seen++;
        }
        if (active == 0 && S.next_eta >= S.neta && fin_slots.empty() && restart.empty()) {
            if (seen == chunk) finished = true;     // nothing running, nothing queued
            return SCINT_OK;                        // else: keep draining the chunks in flight
        }
        return enqueue(fin_chunk);
    }
};

// Shared driver of scint_eval_sweep (eigenvalues), scint_eigvec_sweep (eigenpairs),
// scint_eval_sweep_multi and the chi^2 sweep.  `ncs` conjugate spectra of one shape live
// `cs_stride` elements apart from `cs`, each with its own geometry geom[c] and theta grid
// th_cents + c*M; curvature e reads spectrum cs_index[e] (nullptr: all read spectrum 0).
int32_t run_sweep(const scint_c128* cs, int64_t ncs, int64_t cs_stride, const int32_t* cs_index,
                  const scint_cs_geom* geom, const double* th_cents,
                  int64_t M, const int32_t* keep_idx, const int32_t* keep_n, const double* etas,
                  int64_t neta, double tol, int32_t max_iter, int64_t batch, double* eigs_out,
                  int32_t* status_out, int32_t* iters_out, bool want_vec, cplx* vec_out,
                  int64_t vstride, SweepTail* tail_hook, void* workspace, size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(cs && geom && th_cents && keep_idx && keep_n && etas && eigs_out && status_out && workspace,
                  "sweep: null pointer");
    SCINT_REQUIRE(M >= 1 && neta >= 1 && batch >= 1 && max_iter >= 1 && tol > 0, "sweep: bad arguments");
    SCINT_REQUIRE(ncs >= 1 && (ncs == 1 || cs_index), "sweep: bad spectrum table");
    for (int64_t c = 0; c < ncs; ++c) {
        SCINT_REQUIRE(geom[c].dtau > 0 && geom[c].dfd > 0, "sweep: tau and fd must be increasing");
        SCINT_REQUIRE(geom[c].ntau == geom[0].ntau && geom[c].nfd == geom[0].nfd,
                      "sweep: all conjugate spectra must have one shape");
        SCINT_REQUIRE((int64_t)geom[c].ntau * (int64_t)geom[c].nfd < ((int64_t)1 << 31),
                      "sweep: a conjugate spectrum of 2^31 elements or more (the packed gather indexes it with 32 bits)");
    }
    SCINT_REQUIRE(!want_vec || (vec_out && vstride >= M), "sweep: bad eigenvector output");
    hipStream_t stream = (hipStream_t)stream_;
    const int mode = sweep_mode();              // the ONE read of the process-wide switch for this sweep
    size_t need = 0;
    sweep_workspace_bytes_mode(M, neta, batch, max_iter, want_vec, ncs, mode, &need);
    if (workspace_bytes < need) { set_error("scint: sweep workspace too small"); return SCINT_E_WORKSPACE; }
    SideStreams* side = side_streams();
    if (!side) { set_error("scint: could not create the internal sweep streams and events"); return SCINT_E_HIP; }
    {   // dynamic-LDS limits of the mat-vec kernels: once per (thread, device)
        thread_local std::map<int, bool> lds_set;
        int dev = 0;
        SCINT_HIP(hipGetDevice(&dev));
        if (!lds_set[dev]) {
            SCINT_HIP(hipFuncSetAttribute((const void*)pk2_matvec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kMatvecLdsBytes));
            SCINT_HIP(hipFuncSetAttribute((const void*)pk2_matvec32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kMatvec32LdsBytes));
            SCINT_HIP(hipFuncSetAttribute((const void*)pk2_matvec_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kMatvecMixedLdsBytes));
            lds_set[dev] = true;
        }
    }

    sweep_stats() = SweepStats();
    SweepProblem S;
    S.cs = (const cplx*)cs; S.cs_stride = cs_stride; S.cs_index = cs_index; S.th_cents = th_cents; S.M = M;
    S.keep_idx = keep_idx; S.keep_n = keep_n; S.etas = etas; S.neta = neta;
    S.eigs_out = eigs_out; S.status_out = status_out; S.iters_out = iters_out;
    S.want_vec = want_vec; S.vec_out = vec_out; S.vstride = vstride; S.tail_hook = tail_hook;
    S.mixed = mode_is_mixed(mode, want_vec); S.tol = tol;
    for (int l = 0; l < kTailLanes; ++l) S.tail[l] = side->tail[l];
    S.nbmax = (int)ceil_div(M, kTB);
    S.steps_cap = (int)std::min<int64_t>(std::min<int64_t>(max_iter, M), kMaxK);
    const int nslots = (int)std::min(batch, neta);
    const SweepSchedule& sched = sweep_schedule();
    const int forced_depth = sched.depth.load(), forced_every = sched.check_every.load(), forced_groups = sched.groups.load();
    S.depth = forced_depth >= 1 && forced_depth <= 2 ? forced_depth : 2;
    S.check_every = forced_every >= 1 && forced_every <= 16 ? forced_every : kCheckEveryBlock;
    const int ngroups = (nslots >= 4 && forced_groups != 1) ? 2 : 1;
    S.BL = batch_layout(S.nbmax, S.steps_cap, nslots, want_vec, ncs, S.mixed);
    const SlabLayout& L = S.BL.slab;
    S.base = (char*)workspace;
    S.states_dev = (int32_t*)(S.base + S.BL.states);
    S.geoms_dev = (const GeomDev*)(S.base + S.BL.geoms);
    S.scales_dev = (const double*)(S.base + S.BL.scales);

    // pinned staging: geometry table + per group kTabs x {jobs, strips, fresh, fin, fin_eta, row_strip0, flags}
    auto per_tab = [&](size_t nsl) {
        return align_up(sizeof(PackedJob) * nsl, 64) + align_up(sizeof(Strip) * nsl * (size_t)S.BL.smax, 64) +
               (S.mixed ? align_up(sizeof(Strip32) * nsl * (size_t)S.BL.smax, 64) : 0) +
               3 * align_up(sizeof(int32_t) * nsl, 64) + align_up(sizeof(int64_t) * nsl, 64) +
               align_up(sizeof(int32_t) * nsl * (size_t)(S.nbmax + 1), 64) + align_up(sizeof(int32_t) * 4 * nsl, 64);
    };
    const size_t geom_bytes = align_up(sizeof(GeomDev) * (size_t)ncs, 64);
    char* pin = pinned_staging(geom_bytes + per_tab((size_t)nslots) * kTabs + 64 * 9 * kTabs * 2);
    if (!pin) return SCINT_E_HIP;
    GeomDev* h_geoms = (GeomDev*)pin;
    for (int64_t c = 0; c < ncs; ++c) h_geoms[c] = to_dev(geom[c]);
    char* q = pin + geom_bytes;
    auto take = [&](size_t bytes) { char* r = q; q += align_up(bytes, 64); return r; };

    hipEvent_t start_ev = side->start_ev;
    hipError_t he = hipSuccess;
    int32_t rc = SCINT_OK;
    SweepGroup G[2];
    for (int g = 0; g < ngroups; ++g) {
        SweepGroup& grp = G[g];
        grp.P = &S;
        grp.slot0 = g == 0 ? 0 : nslots / 2;
        grp.nslots = (ngroups == 1 ? nslots : (g == 0 ? nslots / 2 : nslots - nslots / 2));
        grp.stream = g == 0 ? stream : side->aux;
        grp.check_stream = side->chk[g];
        const size_t nsl = (size_t)grp.nslots;
        for (int t = 0; t < kTabs; ++t) {
            grp.chunk_done[t] = side->chunk_done[g][t]; grp.export_done[t] = side->export_done[g][t];
            grp.steps_done[t] = side->steps_done[g][t];
            grp.h_jobs[t] = (PackedJob*)take(sizeof(PackedJob) * nsl);
            grp.h_strips[t] = (Strip*)take(sizeof(Strip) * nsl * (size_t)S.BL.smax);
            grp.h_fresh[t] = (int32_t*)take(sizeof(int32_t) * nsl);
            grp.h_fin[t] = (int32_t*)take(sizeof(int32_t) * nsl);
            grp.h_fin_eta[t] = (int64_t*)take(sizeof(int64_t) * nsl);
            grp.h_rs[t] = (int32_t*)take(sizeof(int32_t) * nsl * (size_t)(S.nbmax + 1));
            grp.h_flags[t] = (int32_t*)take(sizeof(int32_t) * 4 * nsl);
            grp.h_strips32[t] = (Strip32*)take(S.mixed ? sizeof(Strip32) * nsl * (size_t)S.BL.smax : 0);
            grp.h_restart[t] = (int32_t*)take(sizeof(int32_t) * nsl);
        }
        grp.jobs.assign(nsl, PackedJob());
        grp.slot_eta.assign(nsl, -1);
        grp.slot_gen.assign(nsl, 0);
        grp.slot_phase.assign(nsl, 0);
        for (int s = 0; s < grp.nslots; ++s) {                    // static part of every slot
            char* sl = S.base + L.total * (size_t)(grp.slot0 + s);
            PackedJob& J = grp.jobs[(size_t)s];
            J.tiles = (cplx*)(sl + L.tiles);
            J.tiles32 = S.mixed ? (c32*)(sl + L.tiles32) : nullptr;
            J.scale32 = nullptr; J.use32 = 0; J.certify = 0; J.iters_base = 0; J.rowgroup_lg = kRows64Lg;
            J.U[0] = (cplx*)(sl + L.U0); J.U[1] = (cplx*)(sl + L.U1);
            J.Q = (cplx*)(sl + L.Q); J.qstride = (int64_t)S.nbmax * kTB; J.qslots = L.qslots;
            J.want_vec = want_vec ? 1 : 0; J.svec = (double*)(sl + L.svec);
            J.rowpart = (cplx*)(sl + L.rowpart); J.colpart = (cplx*)(sl + L.colpart);
            J.row_strip0 = nullptr;       // set per table copy (SweepGroup::enqueue)
            J.apart[0] = (double*)(sl + L.apart0); J.apart[1] = (double*)(sl + L.apart1);
            J.upart[0] = (double*)(sl + L.upart0); J.upart[1] = (double*)(sl + L.upart1);
            J.coef = (double*)(sl + L.coef);
            J.alpha = (double*)(sl + L.alpha); J.beta = (double*)(sl + L.beta);
            J.result = (double*)(sl + L.result); J.state = S.states_dev + 4 * (grp.slot0 + s);
            J.tol = tol; J.gen = 0;
            J.vec_gap_factor = tail_hook ? kVecGapFactorChisq : kVecGapFactor;     // (packed.hpp: the chi^2 sweep's product is a scalar)
            J.n = 0; J.nb = 1; J.max_steps = 0; J.start = 0; J.strip_len = 1;
            J.eta = 0; J.two_eta = 0; J.keep = keep_idx;
            J.cs = (const cplx*)cs; J.th = th_cents; J.geom = 0; J.pad1 = 0;
            J.eig_out = eigs_out; J.status_out = status_out; J.iters_out = iters_out;
        }
    }
    // State words and geometry table on the caller's stream; the internal streams start after
    // that (and after whatever the caller queued before us: the conjugate spectrum).
    if (rc == SCINT_OK) {
        he = hipMemcpyAsync((void*)S.geoms_dev, h_geoms, sizeof(GeomDev) * (size_t)ncs, hipMemcpyHostToDevice, stream);
        if (he == hipSuccess) he = hipMemsetAsync(S.states_dev, 0, sizeof(int32_t) * 4 * (size_t)nslots, stream);
        if (he == hipSuccess && S.mixed) {
            const int32_t src = launch_cs_scale((const cplx*)cs, ncs, cs_stride, (int64_t)geom[0].ntau * geom[0].nfd,
                                                (unsigned long long*)(S.base + S.BL.scale_bits), (double*)(S.base + S.BL.scales), stream);
            if (src != SCINT_OK) rc = src;
        }
        // the caller's output buffers need no preparation: a status that is never written reads as a failure
        // (0x7f7f7f7f), step counts as 0, eigenvector rows are zero beyond their N_i entries
        if (he == hipSuccess) he = hipMemsetAsync(status_out, 0x7f, sizeof(int32_t) * (size_t)neta, stream);
        if (he == hipSuccess && iters_out) he = hipMemsetAsync(iters_out, 0, sizeof(int32_t) * (size_t)neta, stream);
        if (he == hipSuccess && want_vec)
            he = hipMemsetAsync(vec_out, 0, sizeof(cplx) * (size_t)neta * (size_t)vstride, stream);
        if (he == hipSuccess) he = hipEventRecord(start_ev, stream);
        for (int l = 0; l < kTailLanes && he == hipSuccess; ++l) he = hipStreamWaitEvent(side->tail[l], start_ev, 0);
        if (he == hipSuccess) he = hipStreamWaitEvent(side->aux, start_ev, 0);
        if (he != hipSuccess) rc = hip_fail(he, "sweep setup", __FILE__, __LINE__);
    }
    // Two groups that start together stay in step: equal chunks of equal work, so their convergence checks (one
    // wavefront per curvature, 160 us), refill gathers and reductions coincide and the GPU has no mat-vec to run
    // beside them (7 + 6 ms of a 190 ms sweep in the round-3 trace, profiles/r03_timeline.txt).  The second group
    // therefore waits for the first one's first pass.  Nothing per-job depends on it.
    if (rc == SCINT_OK && ngroups == 2) S.stagger_ev = side->stagger_ev;
    bool staggered = false;
    while (rc == SCINT_OK) {
        bool any = false;
        for (int g = 0; g < ngroups && rc == SCINT_OK; ++g) {
            if (G[g].finished) continue;
            any = true;
            if (g == 1 && !staggered) {
                staggered = true;
                if (G[0].chunk > 0 && G[0].nstrips + G[0].nstrips32 > 0) {
                    he = hipStreamWaitEvent(G[1].stream, S.stagger_ev, 0);
                    if (he != hipSuccess) { rc = hip_fail(he, "sweep stagger", __FILE__, __LINE__); break; }
                }
            }
            rc = G[g].advance();
        }
        if (!any) break;
    }
    // leave nothing running on the internal streams, and nothing pending on the caller's; an
    // asynchronous fault in the last queued chunk or in a tail step surfaces here and nowhere else
    hipError_t sync_err = hipStreamSynchronize(stream);
    { const hipError_t e2 = hipStreamSynchronize(side->aux); if (sync_err == hipSuccess) sync_err = e2; }
    for (auto& c : side->chk) { const hipError_t e2 = hipStreamSynchronize(c); if (sync_err == hipSuccess) sync_err = e2; }
    for (int l = 0; l < kTailLanes; ++l) {
        const hipError_t e2 = hipStreamSynchronize(side->tail[l]);
        if (sync_err == hipSuccess) sync_err = e2;
    }
    { const hipError_t e2 = hipGetLastError(); if (sync_err == hipSuccess) sync_err = e2; }
    if (rc == SCINT_OK && sync_err != hipSuccess) rc = hip_fail(sync_err, "sweep completion", __FILE__, __LINE__);
    return rc;      // (streams and events stay with the thread: side_streams())
}

}  // namespace scint

using namespace scint;

extern "C" int32_t scint_sweep_precision(int32_t mode) {
    std::atomic<int>& m = sweep_mode_ref();
    if (mode >= 0 && mode <= 2) return m.exchange(mode);
    if (mode != -1) { set_error("scint: sweep_precision: mode must be 0 (f64), 1 (mixed), 2 (mixed-all) or -1 (query)"); return SCINT_E_ARG; }
    return m.load();
}

extern "C" int32_t scint_sweep_schedule(int32_t depth, int32_t check_every, int32_t groups) {
    SCINT_REQUIRE(depth >= -1 && depth <= 2 && check_every >= -1 && check_every <= 16 && groups >= -1 && groups <= 2,
                  "sweep_schedule: depth in 0..2, check_every in 0..16, groups in 0..2 (0 = default, -1 = leave as is)");
    SweepSchedule& sc = sweep_schedule();
    if (depth >= 0) sc.depth = depth;
    if (check_every >= 0) sc.check_every = check_every;
    if (groups >= 0) sc.groups = groups;
    return SCINT_OK;
}

extern "C" int32_t scint_sweep_workgroups(int32_t nb, int32_t complex64) {
    if (nb < 1) { set_error("scint: sweep_workgroups: nb must be >= 1"); return -SCINT_E_ARG; }   // (a count is returned: errors are negative)
    const int R = complex64 ? kRows32 : kRows64, S = complex64 ? strip_len32_for(nb) : strip_len_for(nb);
    int n = 0;
    for (int I = 0; I < nb; I += R) n += row_strip_count(nb, I, S, R);     // the strips SweepGroup::enqueue builds for one job
    return n;
}

extern "C" int32_t scint_sweep_stats(double* out) {
    SCINT_REQUIRE(out, "sweep_stats: null pointer");
    const SweepStats& st = sweep_stats();
    out[0] = st.bytes32; out[1] = st.bytes64; out[2] = st.certified; out[3] = st.cert_passes;
    return SCINT_OK;
}

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
                     eigs_out, status_out, iters_out, false, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
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
                     max_iter, batch, eigs_out, status_out, iters_out, false, nullptr, 0, nullptr, workspace,
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
                     w_out, status_out, iters_out, true, (cplx*)vec_out, vec_stride, nullptr, workspace, workspace_bytes,
                     stream);
}
extern "C" int32_t scint_eigvec_sweep_multi_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                                            int32_t max_iter, int64_t ncs, size_t* bytes) {
    return sweep_workspace_bytes(M, neta, batch, max_iter, true, ncs, bytes);
}

extern "C" int32_t scint_eigvec_sweep_multi(const scint_c128* cs_stack, int64_t ncs, int64_t cs_stride,
                                            const int32_t* cs_index, const scint_cs_geom* geoms,
                                            const double* th_stack, int64_t M, const int32_t* keep_idx,
                                            const int32_t* keep_n, const double* etas, int64_t neta,
                                            double tol, int32_t max_iter, int64_t batch, double* w_out,
                                            scint_c128* vec_out, int64_t vec_stride, int32_t* status_out,
                                            int32_t* iters_out, void* workspace, size_t workspace_bytes,
                                            void* stream) {
    SCINT_REQUIRE(cs_index != nullptr || ncs == 1, "eigvec_sweep_multi: cs_index required");
    return run_sweep(cs_stack, ncs, cs_stride, cs_index, geoms, th_stack, M, keep_idx, keep_n, etas, neta, tol,
                     max_iter, batch, w_out, status_out, iters_out, true, (cplx*)vec_out, vec_stride, nullptr,
                     workspace, workspace_bytes, stream);
}
