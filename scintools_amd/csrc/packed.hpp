// packed.hpp -- Hermitian tile-packed theta-theta storage used by the eta sweep.
//
// The sweep never returns the matrix, and theta-theta is Hermitian with a zero diagonal
// (ththmod.py:108-114), so only the tiles on or above the block diagonal are stored:
//
//   n  x n matrix -> nb = ceil(n/64) block rows; tile (I, J), J >= I, is a dense
//   row-major 64 x 64 complex128 block (64 KiB, contiguous) at tile index
//   tile_offset(nb, I) + (J - I); rows/cols >= n are zero.  Diagonal tiles hold the full
//   Hermitian block (both triangles) so that one code path multiplies every tile.
//
// This halves the HBM bytes of both the gather (writes) and every Lanczos mat-vec
// (reads): 8 N^2 instead of 16 N^2, and at N = 4095 the whole matrix (136 MB) fits the
// 256 MiB Infinity Cache.  y = A x is computed tile by tile -- y_I += A_IJ x_J and, for
// J > I, y_J += A_IJ^H x_I -- into per-strip / per-tile partial vectors that a second
// kernel sums in a fixed order, so the result is bit-reproducible (no atomics).
#pragma once
#include <stdlib.h>

#include "thth.hpp"

namespace scint {

constexpr int kTB = 64;                       // tile edge
constexpr int kTileElems = kTB * kTB;         // 4096 complex = 64 KiB

// complex64 copy of a tile (32 KiB): what the ITERATION of the mixed-precision sweep streams (eigen_packed.hip,
// "f32-stored iteration, f64 certificate"); the eigenvalue that is returned never comes from it
struct __attribute__((aligned(8))) c32 { float x, y; };
static_assert(sizeof(c32) == 8, "c32 must be two floats");
#ifndef SCINT_ROWS32
#define SCINT_ROWS32 8            // (round 5, call 1, interleaved: 8 rows x <= 12 tiles 2399 / 2400 eta/s against 2353 / 2350 with 4 x <= 14 --
#endif                            //  the step that gave the complex128 kernel its last 2 % gives the complex64 one the same)
constexpr int kRows32 = SCINT_ROWS32;   // block rows per workgroup of the complex64 mat-vec
constexpr int kRows32Lg = kRows32 == 8 ? 3 : (kRows32 == 4 ? 2 : (kRows32 == 2 ? 1 : 0));
static_assert((1 << kRows32Lg) == kRows32, "kRows32 must be 1, 2, 4 or 8");
#ifndef SCINT_MAXSTRIP32
#define SCINT_MAXSTRIP32 (SCINT_ROWS32 == 8 ? 12 : 14)
#endif
constexpr int kMaxStrip32 = SCINT_MAXSTRIP32;   // its column tiles per strip: X_J blocks and column partials in LDS, 72 KiB in all (14 with 4 rows, 12 with 8)

__host__ __device__ inline int64_t tile_offset(int nb, int I) {
    return (int64_t)I * nb - (int64_t)I * (I - 1) / 2;
}
__host__ __device__ inline int64_t tile_count(int nb) { return (int64_t)nb * (nb + 1) / 2; }

// strips: consecutive tiles (I, J0..J1) of one block row handled by one workgroup
// (long strips amortise the per-workgroup prologue/epilogue; the length depends on nb only, so
// the summation order -- and every bit of the result -- is independent of the batch)
#ifndef SCINT_ROWS64
#define SCINT_ROWS64 8          // (a build constant for the A/Bs of round 4 only: tools/build_variant.sh -DSCINT_ROWS64=2|4|16)
#endif
constexpr int kRows64 = SCINT_ROWS64;     // block rows per workgroup of the complex128 mat-vec: they share the X_J blocks and ONE column
                                          // partial per column tile.  Round 3: 2 rows x <= 16 tiles; round 4: 4 x <= 14 (18 partial vectors per 56
                                          // tiles instead of per 32: 1435 -> 1494 eta/s), then 8 x <= 12 (20 per 96: +1.1-1.4 % on two boxes, mat-vec
                                          // 5.63 -> 5.74 TB/s in the sweep; 16 rows x 8 / 6 tiles: no better / worse -- profiles/r04_tail_schedule_ab.txt)
constexpr int kRows64Lg = kRows64 == 16 ? 4 : (kRows64 == 8 ? 3 : (kRows64 == 4 ? 2 : 1));
static_assert(kRows64 == 2 || kRows64 == 4 || kRows64 == 8 || kRows64 == 16, "kRows64 must be 2, 4, 8 or 16");
#ifndef SCINT_MAXSTRIP
#define SCINT_MAXSTRIP (SCINT_ROWS64 == 8 ? 12 : (SCINT_ROWS64 == 4 ? 14 : (SCINT_ROWS64 == 16 ? 8 : 16)))
#endif
constexpr int kMaxStrip = SCINT_MAXSTRIP;            // column tiles per mat-vec workgroup (their X_J blocks and column partials live in LDS:
                                                     // 8 rows x 12 tiles -> 72 KiB, two workgroups per CU and 12 KiB left for the reduce blocks;
                                                     // 4 x 14 -> 72 KiB; 4 x 9 -> 52 KiB, three per CU; 4 x 6 -> 40 KiB, four: both measured slower)
// Eigenvector sweeps stop when the Ritz residual is below `vec_gap_factor * tol` of the spectral gap theta_1 - theta_2
// (pk2_check_kernel): the angle between the Ritz vector and the eigenvector is then <= vec_gap_factor * tol.
//   kVecGapFactor       30 (3e-11): eigvec sweeps whose VECTOR is the product -- modeler and the phase retrieval, where the
//                       chunks' wavefields are stitched by their relative phases and the mosaic is compared with the reference's
//                       at 1e-9 (measured 1.3e-10; with 100 the GPU suite measured 1.16e-9 on the tutorial mosaic: round 5, call 1);
//   kVecGapFactorChisq  100 (1e-10): the chi^2 sweep, whose product is the scalar chi^2 (every chi^2 parity test holds its 1e-9
//                       with it: GPU suite of the same call); 829 -> 867 (batched tail) -> 878 eta/s with it, 896 with 300 (not taken:
//                       tools/experiments/vec_factor_passes.py shows model deviations of 6e-11 there, a 3x margin only).
#ifndef SCINT_VEC_GAP_FACTOR
#define SCINT_VEC_GAP_FACTOR 30.0
#endif
#ifndef SCINT_VEC_GAP_FACTOR_CHISQ
#define SCINT_VEC_GAP_FACTOR_CHISQ 100.0
#endif
constexpr double kVecGapFactor = SCINT_VEC_GAP_FACTOR;
constexpr double kVecGapFactorChisq = SCINT_VEC_GAP_FACTOR_CHISQ;

inline int strip_len_for(int nb) {
    static const int forced = [] { const char* e = getenv("SCINT_STRIP_LEN"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced > kMaxStrip ? kMaxStrip : forced;   // tests of schedule independence
    // (16 <= nb < 32, N = 961 .. 1984 -- the chunks of Dynspec.fit_thetatheta with cwf = cwt = 256, npad = 3: 8 tiles until round 5; measured on
    //  bench.py --workload fit_thetatheta, 35 840 (chunk, eta) jobs at nb = 19: 4 / 6 / 8 / 12 tiles -> 1.675 / 1.734 / 1.599 / 1.558 s per fit)
    return nb >= 16 ? kMaxStrip : (nb >= 8 ? 4 : (nb >= 4 ? 2 : 1));
}

struct PackedJob {
    // ---- gather ----------------------------------------------------------------
    double eta, two_eta;
    const cplx* cs;         // conjugate spectrum this job reads (one per chunk/observation)
    const double* th;       // [M] theta centres of that chunk
    int32_t geom, pad1;     // index into the launch's GeomDev table
    const int32_t* keep;    // [n] indices into th
    int32_t n, nb;
    cplx* tiles;            // [tile_count(nb)][64][64]
    c32* tiles32;           // the same tiles as complex64, scaled by *scale32 (nullptr: not wanted)
    const double* scale32;  // power of two that brings max |CS| of this job's spectrum into [0.5, 1)
    int32_t use32;          // 1: the mat-vec of this job streams tiles32 (iteration phase of the mixed sweep)
    int32_t certify;        // 1: restarted on the complex128 tiles from the Ritz vectors of the iteration phase
    int32_t iters_base;     // passes of the iteration phase (certify: added to the reported count; restart: its T_k has 2 iters_base rows)
    int32_t rowgroup_lg;    // log2 of the block rows that share one column partial: kRows64Lg (complex128 strips), kRows32Lg (complex64)
    // ---- Lanczos state -----------------------------------------------------------
    int32_t max_steps, strip_len;
    int32_t start, gen;     // launch index of this job's Lanczos step 0; generation of the slot (>= 1)
    cplx* U[2];             // u_{j-1} / u_j                     [nb*64] each
    cplx* Q;                // q vectors: slot (j % qslots) holds q_j, slots are qstride apart
    int64_t qstride;        // elements between slots (>= nb*64)
    int32_t qslots;         // 2 = ring (eigenvalue only); max_steps+1 = keep all (Ritz vector wanted)
    int32_t want_vec;       // 1: stop on the Ritz residual and export the eigenvector of T_k
    double* svec;           // eigenvector of T_k for theta_1 (want_vec / mixed hand-over) and, kSvecStride complex elements on, for theta_2 (hand-over)
    cplx* rowpart;          // [nstrips][64]   row-block partial sums per strip
    cplx* colpart;          // [ntiles][64]    column-block partial sums per off-diagonal tile
    const int32_t* row_strip0;  // [nb+1] first strip index of each block row
    double* apart[2];       // [nb] partial q_j^H u_j      (ping-pong by step parity)
    double* upart[2];       // [nb] partial |u_j|^2
    double* coef;           // [2][16] the step's 2x2 coefficients A_{j-1}, B_{j-1}, 1/diag(B_{j-1}) (by step parity)
    double* alpha;          // [max_steps + 1]
    double* beta;           // [max_steps + 2]   beta[i] couples i-1 and i
    double* result;         // [4] theta, err estimate, resid, theta2
    int32_t* state;         // [4] last FINISHED generation of the slot (job done <=> state[0] >= gen), steps, 1 = the iteration
                            //     phase of a mixed sweep has converged and hands over to a certificate run (the host restarts the slot), -
    double* eig_out; int32_t* status_out; int32_t* iters_out;
    double tol;             // target relative accuracy of the eigenvalue
    double vec_gap_factor;  // want_vec: stop when the Ritz residual <= vec_gap_factor * tol * (theta_1 - theta_2)
};

// One mat-vec workgroup's work: the tiles (I + r, J0 .. J0+ntile-1) of the block rows I + r, r = 0 .. nrows-1 <= kRows64
// (row I + r starts at column max(J0, I + r): nothing below the diagonal is stored).  The rows share the blocks X_J in
// LDS and ONE column partial per column tile (stored in the slots of row I's tiles): a byte of partial-vector traffic
// costs four times a byte read (profiles/r03_pk2e_probe.txt).  The record carries every pointer the workgroup needs, so
// that its tile loads are issued after ONE dependent load (this record) instead of three (strip -> job table -> state word).
struct __attribute__((aligned(16))) Strip {
    const cplx* tiles[kRows64];   // first tile of row I + r in the strip (column max(J0, I + r))
    cplx* rowpart[kRows64];       // row I + r's [64][2] row partials of this strip
    const cplx* Q;                // the job's Q ring (slot j % qslots holds Q_j, slots qstride*2 elements apart)
    cplx* colpart;                // [ntile][64][2] column partials, first column tile of the strip first
    const int32_t* state;         // the slot's state word (job done <=> state[0] >= gen)
    int64_t qstride;
    int32_t I, J0, ntile, qslots;
    int32_t start, gen, max_steps, nrows;
};

// Rows are grouped (0..R-1), (R..2R-1), ...; a short last group runs with the rows it has.  Every row of a group is
// cut on the FIRST row's column grid, so all have the same number of strips.
inline int row_strip_count(int nb, int I, int S, int R) {
    const int lead = I - I % R;                        // the row whose grid this row is cut on
    return (nb - lead + S - 1) / S;
}

// One workgroup of the complex64 mat-vec: tiles (I + r, J0 .. J0+ntile-1) for r = 0 .. nrows-1 (row I + r starts at
// column max(J0, I + r): nothing below the diagonal is stored).  One column partial per column tile for the group.
struct __attribute__((aligned(16))) Strip32 {
    const c32* tiles[kRows32];  // first tile of row I + r in the strip
    cplx* rowpart[kRows32];     // row I + r's [64][2] row partials of this strip
    const cplx* Q;
    cplx* colpart;              // [ntile][64][2]
    const int32_t* state;
    int64_t qstride;
    int32_t I, J0, ntile, qslots;
    int32_t start, gen, max_steps, nrows;
};

// Gather for the jobs in slots[0..njobs) (device array of indices into jobs_dev); every job
// names its own CS, theta grid and geometry (geoms_dev[job.geom]).
// with32: every job also gets its complex64 copy (job.tiles32, scaled by *job.scale32).
int32_t launch_gather_packed(const GeomDev* geoms_dev, int64_t M, const PackedJob* jobs_dev,
                             const int32_t* slots_dev, int njobs, int nbmax, hipStream_t stream, bool with32 = false);
// scale[c] = 2^-e with max(|re|, |im|) over the finite elements of spectrum c in [2^(e-1), 2^e) (1 when that
// maximum is 0); `bits` is scratch of ncs words.  Queued on `stream`.
int32_t launch_cs_scale(const cplx* cs, int64_t ncs, int64_t cs_stride, int64_t nelem, unsigned long long* bits,
                        double* scale, hipStream_t stream);

// What the sweep does with a curvature once its eigenpair has been exported (chi^2 sweep): called
// on the host while the sweep runs; enqueues on a tail stream that already waits for the export.
#ifndef SCINT_TAIL_LANES
#define SCINT_TAIL_LANES 2        // (round 5, call 2: 2 lanes 870 eta/s against 857 / 861 with 4 on the chi^2 objective: batches of <= 8 curvatures keep a lane busy)
#endif
constexpr int kTailLanes = SCINT_TAIL_LANES;   // tail streams of the chi^2 sweep (model steps of retired curvatures in flight at once)
struct SweepTail {
    // `lane` (0 .. kTailLanes-1) names the tail stream: work of one lane is ordered, the lanes overlap, so
    // an implementation keeps one set of scratch buffers per lane
    virtual int32_t retire(int64_t eta_index, hipStream_t tail, int lane) = 0;
    // The curvatures ONE chunk retired, at most batch_max() of them per call (round 5: a chunk retires 20-30 curvatures at once,
    // and a tail that costs the host a dozen API calls per curvature starves both slot groups of their next chunk).  Default: one by one.
    virtual int32_t retire_batch(const int64_t* eta_index, int n, hipStream_t tail, int lane) {
        for (int k = 0; k < n; ++k) {
            const int32_t rc = retire(eta_index[k], tail, lane);
            if (rc != 0) return rc;
        }
        return 0;
    }
    virtual int batch_max() const { return 1; }
    virtual ~SweepTail() {}
};

// The batched Lanczos sweep (eigen_packed.hip).  `ncs` conjugate spectra of one shape live
// `cs_stride` elements apart from `cs`, each with its own geometry geom[c] and theta grid
// th_cents + c*M; curvature e reads spectrum cs_index[e] (nullptr: all read spectrum 0).
int32_t run_sweep(const scint_c128* cs, int64_t ncs, int64_t cs_stride, const int32_t* cs_index,
                  const scint_cs_geom* geom, const double* th_cents, int64_t M, const int32_t* keep_idx,
                  const int32_t* keep_n, const double* etas, int64_t neta, double tol, int32_t max_iter,
                  int64_t batch, double* eigs_out, int32_t* status_out, int32_t* iters_out, bool want_vec,
                  cplx* vec_out, int64_t vstride, SweepTail* tail_hook, void* workspace, size_t workspace_bytes,
                  void* stream);
int sweep_mode();     // 0 f64, 1 mixed (eigenvalue sweeps), 2 mixed-all (eigenpair sweeps too); scint_sweep_precision(): eigenvalue sweeps iterate on a complex64 copy and certify on the complex128 tiles
int32_t sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch, int32_t max_iter, bool want_vec,
                              int64_t ncs, size_t* bytes);

}  // namespace scint
