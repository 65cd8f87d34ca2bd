/*
 * scint_hip.h -- C ABI of libscint_hip.so: the MI355X (gfx950) implementation of
 * scintools' secondary-spectrum + theta-theta hot path.
 *
 * The reference (danielreardon/scintools) is pure Python and has no FFI; the
 * boundary below is what a ctypes binding inside the reference would call in
 * place of the NumPy/SciPy lines cited on every entry point (paths relative to
 * /root/reference/scintools).  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked HOST;
 *   - arrays are C-contiguous; complex data is interleaved (re, im) float64,
 *     i.e. numpy complex128 / HIP double2;
 *   - sizes are int64_t, flags int32_t, the return value is an int32_t status:
 *     0 = ok, >0 = SCINT_E_*; the message is available from scint_last_error();
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued on it; only entry points documented as synchronous
 *     wait for it.  The sweep entry points split the resident curvatures into two
 *     groups, one driven on `stream`, one on an internal stream, and queue each
 *     group's Lanczos steps two chunks ahead of the convergence flags they have
 *     read (so no stream waits for the host); scint_chisq_sweep additionally
 *     drives two internal streams for the model steps of retired curvatures.
 *     Internal streams are per host thread and device.  The sweep entry points
 *     return with every stream drained;
 *   - no function throws, allocates caller-visible memory, or keeps pointers to
 *     caller buffers after it returns.  Persistent library-owned state, all small
 *     and created on first use: a mutex-guarded cache of FFT twiddle tables per
 *     device; an 8 KiB reduction scratch per (device, stream) used by scint_mean /
 *     scint_chisq; per host thread a pinned flag buffer (4 int32 per resident
 *     curvature) and staging for the sweep's job tables, and, per device, the internal
 *     streams mentioned above and a note of which kernels' LDS-size attribute has been set.
 */
#ifndef SCINT_HIP_H
#define SCINT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCINT_OK 0
#define SCINT_E_ARG 1        /* bad size / null pointer / unsupported option   */
#define SCINT_E_HIP 2        /* a HIP runtime call failed                      */
#define SCINT_E_WORKSPACE 3  /* workspace too small                            */
#define SCINT_E_NOCONV 4     /* eigen iteration hit max_iter (per-eta status)  */
#define SCINT_E_EMPTY 5      /* reduced theta-theta has < 2 points (per-eta)   */
#define SCINT_E_NONFINITE 6  /* non-finite eigenvalue (per-eta)                */

typedef struct { double re, im; } scint_c128;

/* Geometry of a conjugate spectrum CS[ntau, nfd] and of the theta grid, as the
 * reference derives it inside thth_map (ththmod.py:83-97) -- computed on the
 * host by the wrapper with the reference's exact NumPy expressions so that the
 * floor() in the gather sees bit-identical operands. */
typedef struct {
    int64_t ntau, nfd;     /* CS shape: tau is the slow axis                    */
    double tau0, dtau;     /* tau[0], np.diff(tau).mean()            [us]       */
    double fd0, dfd;       /* fd[0],  np.diff(fd).mean()             [mHz]      */
    double tau_max;        /* abs(tau.max())  (crop, ththmod.py:153)            */
    double fd_max;         /* abs(fd.max())   (crop, ththmod.py:154)            */
    double tau1_step;      /* tau[1]-tau[0]   (rev_map bin edges, :214-216)     */
    double fd1_step;       /* fd[1]-fd[0]     (rev_map bin edges, :211-213)     */
} scint_cs_geom;

/* ---- diagnostics -------------------------------------------------------- */
int32_t scint_version(void);
/* HOST buffer; copies the calling thread's last error text. */
int32_t scint_last_error(char* buf, size_t n);
/* Number of visible HIP devices (0 if none): lets the wrapper fail loudly. */
int32_t scint_device_count(void);

/* Optional per-kernel timing for the benchmark: between begin and end every launch of
 * the theta-theta gather kernel ([0]) and of the eigen mat-vec kernel ([1]) is bracketed
 * by hipEvents on its stream.  end() synchronises the device and returns, per kernel, the length
 * in milliseconds of the UNION of its launch intervals (the sweep drives two streams, so
 * launches may overlap), the plain SUM of the individual launch spans (sum / launches is the
 * average a kernel trace reports) and the launch counts.  HOST arrays of `count` entries; the
 * library fills min(count, 8) of them and never writes beyond `count` (since version 101: the
 * entry point wrote a fixed 2, then 3, entries before): [2] is the complex64 mat-vec of the
 * mixed-precision sweep (scint_sweep_precision), [3] the rank-1 back-map and [4] the model
 * transform + chi^2 of the model steps of scint_chisq_sweep; since version 102 [5], [6], [7] are the
 * three kernels of the two-trip scint_sspec (input copy + sums, strided axis, row transforms with
 * |.|^2 / dB).  Not thread-safe; off by default. */
int32_t scint_profile_begin(void);
int32_t scint_profile_end(double* ms_out /*HOST[count]*/, double* ms_sum_out /*HOST[count]*/,
                          int64_t* launches_out /*HOST[count]*/, int32_t count);

/* ---- secondary spectrum: Dynspec.calc_sspec core (dynspec.py:3665-3721) -- */
/* dyn[nf,nt] -> sec[(halve? nrfft/2 : nrfft), ncfft] in dB, where
 * nrfft/ncfft = 2*nextpow2(nf/nt) (dynspec.py:3677-3678).
 * win_t[nt], win_f[nf]: tapers of scint_utils.get_window (:810-832) or NULL.
 * pd_fd[ncfft], pd_td[nrfft/2]: post-darkening sin^2 vectors (dynspec.py:3706-3711),
 * required iff prewhite.  Workspace: scint_sspec_workspace_bytes(). */
int32_t scint_sspec_workspace_bytes(int64_t nf, int64_t nt, size_t* bytes /*HOST*/);
int32_t scint_sspec(const double* dyn, int64_t nf, int64_t nt,
                    const double* win_t, const double* win_f,
                    int32_t prewhite, int32_t halve,
                    const double* pd_fd, const double* pd_td,
                    double* sec_out, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ---- conjugate spectrum: single_search lines ththmod.py:777-787 ---------- */
/* CS = fftshift(fft2(pad(dspec, right/bottom by npad*shape, pad_value)));
 * rows r with mask_lo <= r < mask_hi are zeroed (the |tau| < tauMask rows).
 * If `incoherent` != 0 the output is abs(CS) + 0j (ththmod.py:801).
 * cs_out[(npad+1)*nf, (npad+1)*nt].  Any sizes (non powers of two go through a
 * chirp-z pass).  Workspace: scint_cs_workspace_bytes(). */
int32_t scint_cs_workspace_bytes(int64_t nf, int64_t nt, int64_t npad, size_t* bytes /*HOST*/);
int32_t scint_cs(const double* dspec, int64_t nf, int64_t nt, int64_t npad,
                 double pad_value, int64_t mask_lo, int64_t mask_hi, int32_t incoherent,
                 scint_c128* cs_out, void* workspace, size_t workspace_bytes, void* stream);

/* mean of a device array (np.mean of the chunk), synchronous, HOST result. */
int32_t scint_mean(const double* x, int64_t n, double* mean_out /*HOST*/, void* stream);

/* ---- CS -> theta-theta: thth_map + thth_redmap (ththmod.py:56-173) ------- */
/* th_cents[M]: bin centres (ththmod.py:83-84).  keep_idx[N]: indices (into
 * th_cents, ascending) of the centres kept by the crop (:153-155); pass the
 * identity for plain thth_map.  thth_out[N,N].  hermitian: ththmod.py:108-114. */
int32_t scint_thth_map(const scint_c128* cs, const scint_cs_geom* geom /*HOST*/,
                       const double* th_cents, int64_t M,
                       const int32_t* keep_idx, int64_t N,
                       double eta, int32_t hermitian,
                       scint_c128* thth_out, void* stream);

/* ---- eta sweep: the Eval_calc loop of single_search (ththmod.py:788-799) -- */
/* For each eta_i: reduced theta-theta (as above, hermitian) -> dominant
 * 'largest algebraic' eigenvalue seeded with the middle row (Eval_calc,
 * ththmod.py:396-401) -> eigs_out[i] = |w|.
 * keep_idx[neta*M] / keep_n[neta]: per-eta crop (host-computed, device arrays).
 * status_out[i] != 0 -> the wrapper stores NaN (ththmod.py:795-799).
 * iters_out[i]: Lanczos steps used (for the roofline accounting); may be NULL.
 * eigs_out/status_out/iters_out are DEVICE arrays and need no initialisation (the library presets status to a
 * failure code and the step counts to 0 on the caller's stream; eigenvector rows of the *_vec entry points are
 * zero beyond their N_i entries, chi^2 of a curvature whose crop leaves nothing stays NaN).
 * NOT asynchronous: like every *_sweep entry point this one drives the caller's stream AND the library's internal
 * streams from a host-side scheduler that reads convergence flags back, and returns with all of them drained -- results
 * are complete (in device memory) on return, and an asynchronous fault of any queued kernel is reported by this call. */
int32_t scint_eval_sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                         int32_t max_iter, size_t* bytes /*HOST*/);
/* Operand precision of the ITERATION of the eigenvalue sweeps (scint_eval_sweep, scint_eval_sweep_multi), per process:
 *   0  float64 throughout (the default): every Lanczos pass streams the complex128 tiles;
 *   1  mixed: the passes stream a complex64 copy of theta-theta (float64 vectors and sums), and the eigenvalue that
 *      is returned is the Ritz value of a certificate pass on the complex128 tiles that satisfies the float64 sweep's
 *      own a-posteriori bound, evaluated in float64 on that matrix -- same tolerance, same status codes
 *      (csrc/eigen_packed.hip, "Mixed precision").  iters_out[i] then counts the passes of both phases (complex64 and
 *      complex128; scint_sweep_stats splits the bytes).  Eigenvector sweeps are not affected.
 *   2  mixed-all: as 1, and the eigenPAIR sweeps (scint_eigvec_sweep*, scint_chisq_sweep) iterate on the complex64 copy too,
 *      to the eigenvalue rule; the run that starts from the two Ritz vectors on the complex128 tiles then continues to
 *      the eigenvector rule of the float64 sweep (residual of the float64 matrix, in float64).
 *  -1  query.
 * Returns the previous mode.  Call it BEFORE the *_workspace_bytes of a sweep: the mixed sweep needs a larger
 * workspace.  The environment variable SCINT_SWEEP_PRECISION=mixed|mixed-all|f64 sets the initial mode. */
int32_t scint_sweep_precision(int32_t mode);
/* Scheduling of the sweeps, per process (the host-side scheduler of eigen_packed.hip; the reference has no counterpart --
 * its loop, ththmod.py:788-799, is sequential): chunks queued ahead of the convergence flags the host has seen (`depth`,
 * 1 or 2), Lanczos passes between convergence checks (`check_every`, 1..16), slot groups on separate streams (`groups`,
 * 1 or 2).  0 restores the measured default, -1 leaves a setting as it is.  Initial values: SCINT_SWEEP_DEPTH /
 * SCINT_CHECK_EVERY / SCINT_SWEEP_GROUPS in the environment, read once.  `depth` and `groups` change no bit of any result;
 * `check_every` decides after which passes the stopping rule is looked at, so a curvature may stop a pass earlier or later
 * -- a value inside the same tolerance, not the same bits (tests/test_gpu_edges.py).  They exist for that test and for
 * bench.py's one-slot-group leg. */
int32_t scint_sweep_schedule(int32_t depth, int32_t check_every, int32_t groups);
/* Diagnostics of the calling thread's last sweep (any of the sweep entry points), for the roofline accounting:
 * out[0] algorithmic bytes of its complex64 passes (4 n (n + 1) each), out[1] of its complex128 passes (8 n (n + 1)),
 * out[2] curvatures that went through a certificate, out[3] complex128 passes those certificates took. */
int32_t scint_sweep_stats(double* out /*HOST[4]*/);
/* Mat-vec workgroups ONE theta-theta matrix of `nb` 64-row blocks contributes to a launch of the sweep (`complex64` != 0: of the
 * complex64 kernel of the mixed sweep) -- block rows per workgroup and column tiles per strip are build constants of the library
 * (csrc/packed.hpp).  For callers that size the `batch` argument of the sweeps: resident curvatures x this = workgroups per
 * launch (scintools_amd/ththmod.py: default_batch).  Returns the count (>= 1), or -SCINT_E_ARG for nb < 1. */
int32_t scint_sweep_workgroups(int32_t nb, int32_t complex64);
/* Environment variables the library reads (each ONCE per process unless said otherwise; none selects another kernel family or
 * changes a result beyond what is said here):
 *   SCINT_SWEEP_PRECISION = f64 | mixed | mixed-all   initial value of scint_sweep_precision();
 *   SCINT_SWEEP_DEPTH, SCINT_CHECK_EVERY, SCINT_SWEEP_GROUPS   initial values of scint_sweep_schedule();
 *   SCINT_STRIP_LEN = n   column tiles per mat-vec workgroup, at most the build's maximum (csrc/packed.hpp): for the test that
 *                         proves a sweep's values do not depend on the strip shape beyond rounding (tests/test_emu_cpu.py).
 *   SCINT_SSPEC_MAXGRID = n   (read per call) an upper bound on the workgroups of calc_sspec's persistent kernels, for the test
 *                         that runs several rows / column pairs per workgroup at small shapes; changes no bit of the result.
 * (SCINT_CHISQ_MODEL and SCINT_SSPEC_GENERIC of round 4 are gone: no test used them; the model route of the chi^2 sweep is
 *  reached with a mask or a non-finite dspec, the generic calc_sspec route with halve = 0 or lengths outside 256..8192.) */
/* Limits common to every sweep entry point below (eval / eigvec / chisq, single and _multi): a conjugate spectrum must hold
 * fewer than 2^31 elements (ntau * nfd; the packed gather indexes it with 32 bits) -- SCINT_E_ARG otherwise, before anything
 * is queued.  scint_thth_map / scint_rev_map / scint_modeler index with 64 bits and have no such limit. */
int32_t scint_eval_sweep(const scint_c128* cs, const scint_cs_geom* geom /*HOST*/,
                         const double* th_cents, int64_t M,
                         const int32_t* keep_idx, const int32_t* keep_n /*HOST*/,
                         const double* etas /*HOST*/, int64_t neta,
                         double tol, int32_t max_iter, int64_t batch,
                         double* eigs_out, int32_t* status_out, int32_t* iters_out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same sweep over MANY conjugate spectra of one shape in a single batched call -- the
 * chunk loop of Dynspec.fit_thetatheta (dynspec.py:1681-1719), where every chunk has its own
 * CS, eta grid and (frequency-scaled) edges.  cs_stack[ncs][ntau][nfd] with cs_stride elements
 * between spectra; geoms[ncs] (HOST); th_stack[ncs][M]; curvature e reads spectrum
 * cs_index[e] (HOST) and keeps keep_idx[e*M ..] of th_stack[cs_index[e]]. */
int32_t scint_eval_sweep_multi_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                               int32_t max_iter, int64_t ncs, size_t* bytes /*HOST*/);
int32_t scint_eval_sweep_multi(const scint_c128* cs_stack, int64_t ncs, int64_t cs_stride,
                               const int32_t* cs_index /*HOST*/, const scint_cs_geom* geoms /*HOST*/,
                               const double* th_stack, int64_t M,
                               const int32_t* keep_idx, const int32_t* keep_n /*HOST*/,
                               const double* etas /*HOST*/, int64_t neta,
                               double tol, int32_t max_iter, int64_t batch,
                               double* eigs_out, int32_t* status_out, int32_t* iters_out,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Same sweep, but returning the eigenpair (signed w and unit eigenvector) of every eta: the
 * eigsh call of modeler (ththmod.py:308) for a whole curvature sweep.  vec_out[neta, vec_stride]
 * (vec_stride >= M; row i holds keep_n[i] entries).  Stops on the Ritz residual (tol). */
int32_t scint_eigvec_sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                           int32_t max_iter, size_t* bytes /*HOST*/);
int32_t scint_eigvec_sweep(const scint_c128* cs, const scint_cs_geom* geom /*HOST*/,
                           const double* th_cents, int64_t M,
                           const int32_t* keep_idx, const int32_t* keep_n /*HOST*/,
                           const double* etas /*HOST*/, int64_t neta,
                           double tol, int32_t max_iter, int64_t batch,
                           double* w_out, scint_c128* vec_out, int64_t vec_stride,
                           int32_t* status_out, int32_t* iters_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The crop of thth_redmap (ththmod.py:153-155) for every curvature of a sweep, on the device:
 * keep_idx[e*M ..] = ascending indices i with th_cents[i]^2 * etas[e] < tau_max and |th_cents[i]| < fd_half
 * (NumPy's expression, same roundings), keep_n[e] their count -- the tables the sweeps take.  th_cents,
 * keep_idx, keep_n: DEVICE; etas: HOST.  Asynchronous. */
int32_t scint_sweep_keep(const double* th_cents, int64_t M, const double* etas /*HOST*/, int64_t neta,
                         double tau_max, double fd_half, int32_t* keep_idx, int32_t* keep_n, void* stream);

/* The same for MANY conjugate spectra of one shape in one batched sweep (arguments as
 * scint_eval_sweep_multi): the eigenpairs of all chunks of Dynspec.thetatheta_chunks -- one
 * modeler() call per chunk in the reference, dynspec.py:1765-1826 / ththmod.py:1455 -- in ONE call. */
int32_t scint_eigvec_sweep_multi_workspace_bytes(int64_t M, int64_t neta, int64_t batch,
                                                 int32_t max_iter, int64_t ncs, size_t* bytes /*HOST*/);
int32_t scint_eigvec_sweep_multi(const scint_c128* cs_stack, int64_t ncs, int64_t cs_stride,
                                 const int32_t* cs_index /*HOST*/, const scint_cs_geom* geoms /*HOST*/,
                                 const double* th_stack, int64_t M,
                                 const int32_t* keep_idx, const int32_t* keep_n /*HOST*/,
                                 const double* etas /*HOST*/, int64_t neta,
                                 double tol, int32_t max_iter, int64_t batch,
                                 double* w_out, scint_c128* vec_out, int64_t vec_stride,
                                 int32_t* status_out, int32_t* iters_out,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- dominant eigenpair of a given Hermitian matrix (eigsh k=1 'LA') ----- */
/* a[n,n] row-major.  v0[n] start vector or NULL (then a fixed pseudo-random
 * start, as modeler's eigsh call has no v0, ththmod.py:308).  w_out[1],
 * vec_out[n] (unit 2-norm) device outputs; status_out[1], iters_out[1]. */
int32_t scint_eigh_top_workspace_bytes(int64_t n, int32_t max_iter, size_t* bytes /*HOST*/);
int32_t scint_eigh_top(const scint_c128* a, int64_t n, const scint_c128* v0,
                       double tol, int32_t max_iter,
                       double* w_out, scint_c128* vec_out,
                       int32_t* status_out, int32_t* iters_out,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- theta-theta -> CS: rev_map (ththmod.py:176-271) --------------------- */
/* thth[N,N] on centres th_cents[N] (already re-centred) -> recov[ntau,nfd].
 * If `rank1` != 0, thth is not read: thth = |w| V V^H with V = vec[N], w = *w
 * (device), which is modeler's thth2_red (ththmod.py:312-313).
 * Every pixel of recov_out is written (no zero-fill needed).  The per-pixel sums are
 * order-independent: each addend is split on a fixed binary grid so that the float64 LDS
 * accumulations are exact (thth.hip, RevSplit), hence the image is bit-reproducible from run
 * to run like np.histogram2d's.  Workspace: scint_rev_map_workspace_bytes() (a few words).
 * Version 107: the rank-1 Hermitian image (modeler's) on a UNIFORM theta grid -- theta_k = theta_0 + k step to 1e-9 of a
 * step, checked on the device for every call; the centres of linspace edges and any contiguous crop of them are one -- is
 * formed by another kernel (thth.hip, rev_diag_kernel: whole diagonals j = i + s of a Doppler column instead of a search for
 * every theta_i's partners; plain float64 sums in ONE fixed order, so equally bit-reproducible; the same pairs in the same
 * pixels, each pixel within its rounding of the split sums').  After the call word 9 (uint64) of the workspace is 1 if that
 * kernel formed the image, 0 if the general one did.  Environment: SCINT_REV_DIAG=0 keeps every image (also the chi^2 sweep's)
 * on the general kernel. */
int32_t scint_rev_map_workspace_bytes(size_t* bytes /*HOST*/);
int32_t scint_rev_map(const scint_c128* thth, const scint_c128* vec, const double* w,
                      int32_t rank1, const double* th_cents, int64_t N,
                      const scint_cs_geom* geom /*HOST*/, double eta, int32_t hermitian,
                      scint_c128* recov_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- model dynamic spectrum: ifft2(ifftshift(recov)).real (ththmod.py:322-324) ----------
 * recov[ntau, nfd] -> model_out[ntau, nfd].  Computed as a complex-to-real transform of the
 * Hermitian part of ifftshift(recov) when both lengths are powers of two (nfd in 32..8192): an
 * identity for any input, equal to the complex transform's real part up to rounding. */
int32_t scint_model_workspace_bytes(int64_t ntau, int64_t nfd, size_t* bytes /*HOST*/);
int32_t scint_model_from_recov(const scint_c128* recov, int64_t ntau, int64_t nfd,
                               double* model_out, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- complex inverse: scale * ifft2(ifftshift(x))[:crop_rows, :crop_cols] ----------------
 * The wavefield step of single_chunk_retrieval (ththmod.py:1465-1468).  out[crop_rows, crop_cols]. */
int32_t scint_ifft2_shifted(const scint_c128* in, int64_t rows, int64_t cols, double scale,
                            int64_t crop_rows, int64_t crop_cols, scint_c128* out,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- the mosaic of the retrieved chunks (ththmod.py:1492-1554) and the chunks themselves (dynspec.py:1782-1790) ----------
 * Replaces the host loop of `mosaic`: a chunk is rotated onto the sum of the chunks before it and added under its taper, with
 * NumPy's own summation order and product arithmetic restated on the device (csrc/mosaic.hip) so that the result equals the
 * host loop's bit for bit.  The reference's loop is sequential, but chunk (cf, ct) only meets its four predecessors
 * (cf, ct-1), (cf-1, ct-1 .. ct+1): chunks with equal 2 cf + ct are independent and their windows disjoint, so the wrapper walks
 * 2 (ncf - 1) + nct STEPS and hands each step's chunks to one launch -- every pixel still receives its addends in the
 * reference's order.  Per step: scint_mosaic_phase (sums_out[k][2] <- sum over chunk k's window of (E * conj(chunk)) * mask,
 * device memory), mean / numpy.angle / numpy.exp in NumPy on the host, scint_mosaic_add (E[window] += (chunk * mask) *
 * phases[k]).  jobs: device int64 [count][4] = (offset of the window's first element in E, chunk index in the stack `chunks`
 * [.][cwf][cwt], row taper, column taper), count <= 64; tapers rows[4][cwf], cols[4][cwt] (variant 2 * has-a-neighbour-before +
 * has-a-neighbour-after), mask[r][c] = rows[rv][r] * cols[cv][c]; ldE the wavefield's row length; phases: HOST [count][2].
 * numpy_fused, two properties of the HOST's NumPy that the wrapper measures: bit 0 = it multiplies complex arrays with fused
 * multiply-adds (re = fma(ar, br, -(ai bi)): its x86 SIMD loops) rather than the plain expressions; bit 1 = it evaluates
 * chunk_old * conj(chunk_new) with the operands swapped (temporary elision, chunks of 256 KiB and more).
 * The library keeps one small plan per window size in device memory (as it does FFT twiddle tables).
 * scint_chunk_cut: chunks_out[k] = nan_to_num(dyn[r0[k] : r0[k] + cwf, c0[k] : c0[k] + cwt] - nanmean(that window)),
 * pad_out[k] = mean(chunks_out[k]) (the padding value of the conjugate spectrum, ththmod.py:783); r0 / c0 device int32.
 * colmajor: the host array is Fortran-ordered (numpy.copy keeps the order, NumPy's sums then walk a window column by column);
 * `dyn` itself is row-major here either way.
 * Workspace for both: scint_mosaic_workspace_bytes(cwf, cwt, nchunk) (nchunk = count for scint_mosaic_phase). */
int32_t scint_mosaic_workspace_bytes(int64_t cwf, int64_t cwt, int64_t nchunk, size_t* bytes /*HOST*/);
int32_t scint_mosaic_phase(const scint_c128* E, int64_t ldE, const scint_c128* chunks, int64_t cwf, int64_t cwt,
                           const int64_t* jobs, int64_t count, const double* rows, const double* cols, int32_t numpy_fused,
                           void* workspace, size_t workspace_bytes, double* sums_out, void* stream);
int32_t scint_mosaic_add(scint_c128* E, int64_t ldE, const scint_c128* chunks, int64_t cwf, int64_t cwt,
                         const int64_t* jobs, int64_t count, const double* rows, const double* cols, int32_t numpy_fused,
                         const double* phases /*HOST*/, void* stream);
int32_t scint_chunk_cut(const double* dyn, int64_t nf, int64_t nt, const int32_t* r0, const int32_t* c0, int64_t nchunk,
                        int64_t cwf, int64_t cwt, int32_t colmajor, double* chunks_out, double* pad_out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- the per-chunk steps of Dynspec.thetatheta_chunks, a whole group of chunks per call (dynspec.py:1765-1826; the reference
 * maps single_chunk_retrieval, ththmod.py:1390-1476, over the chunks) --------------------------------------------------------
 * scint_cs_batch: scint_cs of every chunk of dstack [n][nf][nt] into cs_stack [n][R][C]; pads (HOST [n]) the padding values,
 * mask_lohi (HOST [n][2]) the masked delay rows of each.  scint_retrieval_tail: per chunk k with keep_n[k] >= 2 the theta-theta
 * of the E field (zeros with row N/2 = rows[k][:N] = conj(V) sqrt(w), ththmod.py:1459-1461), its non-Hermitian back-map on
 * geoms[k] / etas[k] / th_red[k][:N], and scale * ifft2(ifftshift(.))[:nf, :nt] -> out[k] (scint_ifft2_shifted); chunks with
 * keep_n[k] < 2 are left untouched.  The back-map is rev_map's arithmetic (ththmod.py:176-271, hermetian=False) without the N x N
 * matrix: a pixel is (sum of the row's weights that fall in it, in increasing j) x 1 / (number of ALL N^2 pairs that fall in it),
 * and those counts are formed once per CLASS -- class_id (HOST [n]): consecutive chunks with equal ids share theta grid, curvature
 * and axes (the chunks of one frequency row of an observation).  rows [n][M] complex, th_red [n][M], out [n][nf][nt]: device;
 * keep_n, class_id, geoms, etas: HOST. */
int32_t scint_cs_batch(const double* dstack, int64_t n, int64_t nf, int64_t nt, int64_t npad, const double* pads /*HOST*/,
                       const int64_t* mask_lohi /*HOST*/, int32_t incoherent, scint_c128* cs_stack, void* workspace,
                       size_t workspace_bytes, void* stream);
int32_t scint_retrieval_tail_workspace_bytes(int64_t M, int64_t ntau, int64_t nfd, size_t* bytes /*HOST*/);
int32_t scint_retrieval_tail(const scint_c128* rows, const double* th_red, const int32_t* keep_n /*HOST*/, const int32_t* class_id /*HOST*/,
                             const scint_cs_geom* geoms /*HOST*/, const double* etas /*HOST*/, int64_t n, int64_t M, int64_t nf, int64_t nt,
                             double scale, scint_c128* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Gerchberg-Saxton iterations of Dynspec.gerchberg_saxton (dynspec.py:1868-1875) -------
 * wavefield[rows, cols] in place.  Per iteration: fft2, zero the natural-order delay rows
 * zero_lo <= k < zero_hi (the tau < 0 half), ifft2, then where pos != 0 replace the amplitude
 * by amp (= sqrt(dyn)) keeping the phase.  Workspace: scint_gs_workspace_bytes(). */
int32_t scint_gs_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes /*HOST*/);
int32_t scint_gerchberg_saxton(scint_c128* wavefield, int64_t rows, int64_t cols,
                               const double* amp, const uint8_t* pos,
                               int64_t zero_lo, int64_t zero_hi, int32_t niter,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ---- autocovariance: Dynspec.calc_acf(method='direct') (dynspec.py:3780-3797) ------------
 * acf_out[2nf, 2nt] = real(fftshift(ifft2(|fft2(dyn - mean, s=[2nf, 2nt])|^2))), / max if
 * normalise; the mean is subtracted only if subtract_mean (the reference skips it for
 * input_dyn).  dyn must be finite (the reference takes the mean of the valid pixels only). */
int32_t scint_acf_workspace_bytes(int64_t nf, int64_t nt, size_t* bytes /*HOST*/);
int32_t scint_acf(const double* dyn, int64_t nf, int64_t nt, int32_t subtract_mean, int32_t normalise,
                  double* acf_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- chi^2: sum((model[:nf,:nt]-dspec)[mask]**2)/N (ththmod.py:364-367) --- */
/* mask: uint8[nf*nt] or NULL (= isfinite(dspec)).  out: DEVICE double[1].  Asynchronous. */
int32_t scint_chisq(const double* model, int64_t ld_model, const double* dspec,
                    int64_t nf, int64_t nt, const uint8_t* mask, double noise_n,
                    double* out, void* stream);

/* ---- modeler / chisq_calc sweep: [chisq_calc(dspec, CS, tau, fd, eta, edges, N, mask) for eta in etas]
 * (ththmod.py:330-368 over :274-327) in ONE call.  The dominant eigenPAIR of every curvature is found
 * by the batched Lanczos sweep (as scint_eigvec_sweep); as each curvature retires, its model step --
 * rank-1 rev_map of |w| V V^H, inverse FFT, chi^2 against dspec[nf, nt] -- is chained on an internal
 * stream beside the Lanczos steps of the curvatures still resident, without returning to the host.
 * When the model is not cropped (ntau == nf, nfd == nt), mask is NULL and dspec is finite, chi^2 is taken
 * from recov and fft2(dspec) by Parseval's identity instead (the same sum to rounding; no inverse FFT); on that route
 * the curvatures one chunk of the sweep retires go through their model step TOGETHER (<= 8 per set of launches), and
 * back-map and chi^2 touch only the delay rows |tau| <= |eta| max theta^2 a curvature can reach (recov is exactly 0
 * elsewhere; those rows contribute sum |fft2(dspec)|^2).  The workspace holds up to 8 images per internal stream for it.
 *   th_red   DEVICE [neta, M]: row e = the N_e centres of the reduced edges (ththmod.py:157-172 then
 *            :204-205), host-computed like the other grid quantities;
 *   crop_group HOST [neta] or NULL: curvatures with the same non-negative id keep the same theta centres (their th_red rows
 *            are equal element for element: the caller's promise); -1 = on its own.  Which theta_j pair with theta_i in a
 *            Doppler column does not depend on the curvature, so the back-maps of the two largest groups (>= 8 members)
 *            share one partner table per group instead of walking the centres per curvature (since version 103).  Version
 *            107: a curvature whose th_red row is a uniform grid (tested on the device before the sweep: every grid of the
 *            reference's path is one) takes the diagonal back-map of scint_rev_map instead, which has no partners to look
 *            up; tables are built only for groups of curvatures that are not;
 *   mask     DEVICE uint8[nf*nt] or NULL (= isfinite(dspec));
 *   chisq_out DEVICE [neta]: sum((model[:nf,:nt]-dspec)[mask]**2)/noise_n; NaN-filled by the call, and left NaN for a
 *            curvature whose crop keeps fewer than THREE centres (two have no mean edge step: the reference's
 *            rev_map raises there, ththmod.py:166) or whose eigen-solve failed (see status_out);
 *   w_out / vec_out / status_out / iters_out as in scint_eigvec_sweep.
 * Synchronous like the other sweep entry points (returns with all internal streams drained). */
/* Version 107: when, on that route, the axes are symmetric about 0 (even lengths, x0 = -(n / 2) step to 1e-7 of a step: every
 * fft_axis), chi^2 of a uniform-grid curvature comes from the back-map's accumulators -- the rank-1 Hermitian histogram is then
 * mirror-symmetric, fft2(model) at an interior pixel IS recov there, and the workgroup that holds the pixel adds |recov - D|^2
 * itself: the image is neither written nor read back (Doppler column 0 and delay row 0, whose mirrors are off the axes, still
 * are, and take the partner formula).  A pair that sits on a bin edge with its mirrored pair NOT in the mirrored pixel raises the
 * curvature's flag (the edges np.histogram2d computes mirror each other only to ~1e-9 of a step); flagged curvatures are done
 * again from a written image before the call returns.  SCINT_CHISQ_FUSE=0 (environment) keeps every image written.
 * scint_chisq_sweep_last_route: of the LAST scint_chisq_sweep of the process -- fused = 1 if it took chi^2 from the accumulators,
 * redone = the curvatures it did again (tests, bench.py). */
int32_t scint_chisq_sweep_last_route(int32_t* fused /*HOST*/, int64_t* redone /*HOST*/);
int32_t scint_chisq_sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch, int32_t max_iter,
                                          int64_t ntau, int64_t nfd, int64_t nf, int64_t nt,
                                          size_t* bytes /*HOST*/);
int32_t scint_chisq_sweep(const scint_c128* cs, const scint_cs_geom* geom /*HOST*/, const double* th_cents,
                          int64_t M, const int32_t* keep_idx, const int32_t* keep_n /*HOST*/,
                          const double* etas /*HOST*/, int64_t neta, double tol, int32_t max_iter,
                          int64_t batch, const double* th_red, const int32_t* crop_group /*HOST[neta] or NULL*/,
                          const double* dspec, int64_t nf, int64_t nt,
                          const uint8_t* mask, double noise_n, double* chisq_out, double* w_out,
                          scint_c128* vec_out, int64_t vec_stride, int32_t* status_out, int32_t* iters_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ==== arc normalisation: scale_dyn / norm_sspec / fit_arc ==== */

/* ---- equal-wavelength resample: Dynspec.scale_dyn(scale='lambda') (dynspec.py:3948-3957) ----
 * Every time column of dyn[nf, nt] is interpolated with the not-a-knot cubic spline
 * (scipy interp1d(kind='cubic')) from the channel frequencies to nout target frequencies, and
 * the rows are written flipped (np.flipud): out[nout-1-k, :] = S(target k).
 * The spline is solved in second-derivative form (moments M[i]); the tridiagonal system
 * depends on the frequency axis only, so the caller passes its Thomas factors (DEVICE arrays):
 *   h[nf-1] knot spacings; interior rows i = 1..nf-2 with right-hand side
 *   r_i = 6*((y[i+1]-y[i])/h[i] - (y[i]-y[i-1])/h[i-1]):  forward  d_i = (r_i - sub[i]*d_{i-1})*inv[i],
 *   backward M[i] = d_i - sup[i]*M[i+1];  not-a-knot ends (HOST end[4]):
 *   M[0] = end[0]*M[1] + end[1]*M[2],  M[nf-1] = end[2]*M[nf-2] + end[3]*M[nf-3];
 *   per target k: interval idx[k] and coef[k][4]: S = c0*y[i] + c1*y[i+1] + c2*M[i] + c3*M[i+1].
 * `reverse` != 0: the frequency axis is descending and row i of the spline is dyn row nf-1-i.
 * Both sweeps contract, so they run in independent frequency blocks of `block_rows` rows, each
 * started `warm` rows early from zero; the caller sizes `warm` from the factors so that the
 * start-up error is below rounding (block_rows <= 0: one block, the plain sequential sweep).
 * workspace: 2*nf*nt doubles. */
int32_t scint_spline_resample(const double* dyn, int64_t nf, int64_t nt, int32_t reverse,
                              const double* h, const double* sub, const double* inv,
                              const double* sup, const double* end /*HOST [4]*/,
                              int64_t block_rows, int64_t warm,
                              const int32_t* idx, const double* coef, int64_t nout,
                              double* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Dynspec.norm_sspec row loop (dynspec.py:2093-2127) ------------------------------------
 * For delay rows r = 0..nr-1 (row row0+r of sspec[., ld], nc columns, dB):
 *   scale = sqrt(yaxis[row0+r]/eta); sel = |fdop| <= maxnormfac*scale;
 *   norm_out[r, k] = np.interp(x[k], fdop[sel]/scale, sspec[row0+r, sel])   (exact NumPy
 *   branch structure and rounding: searchsorted bracket, slope*(x-xp[j])+fp[j], NaN retry);
 *   mask_out[r, k] = (|x[k]| > max|fdop[sel]/scale|) or isnan(norm_out[r, k]);
 *   pow_out[r]     = mean over unmasked finite entries of 10**(v/10), v = norm_out[r, k] or,
 *                    when xlin is given (logsteps), np.interp(xlin[k], ...)  (NaN if none).
 * Columns [cut_lo, cut_hi) read as NaN (cutmid); row_offset[nr] (or NULL) is subtracted from
 * the row first (subtract_artefacts).  fdop must be ascending.  All pointers DEVICE. */
int32_t scint_norm_sspec(const double* sspec, int64_t ld, int64_t nc, const double* fdop,
                         const double* yaxis, int64_t row0, int64_t nr, double eta,
                         double maxnormfac, int64_t cut_lo, int64_t cut_hi,
                         const double* row_offset, const double* x, const double* xlin,
                         int64_t nx, double* norm_out, uint8_t* mask_out, double* pow_out,
                         void* stream);

/* ---- np.ma.average(norm, axis=0, weights=w) (dynspec.py:2171-2181) --------------------------
 * avg_out[k] = sum_r w[r]*norm[r,k] / sum_r w[r] over rows with rowsel[r] != 0 (NULL = all) and
 * mask[r,k] == 0.  A column with no entry gets empty_out[k] = 1 and avg_out[k] = 0 (the value
 * numpy.ma leaves under the mask).  Fixed summation order (deterministic). */
int32_t scint_masked_colavg_workspace_bytes(int64_t nr, int64_t nx, size_t* bytes /*HOST*/);
int32_t scint_masked_colavg(const double* norm, const uint8_t* mask, int64_t nr, int64_t nx,
                            const double* weights, const uint8_t* rowsel, double* avg_out,
                            uint8_t* empty_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- np.nanmean(sspec[rows, colsel], axis=1) (dynspec.py:2060-2061) ------------------------ */
int32_t scint_row_nanmean(const double* sspec, int64_t ld, int64_t nc, int64_t row0, int64_t nr,
                          const uint8_t* colsel, int64_t cut_lo, int64_t cut_hi, double* out,
                          void* stream);

/* ---- np.std over rows [r0, r1), columns [0, c_lo) + [c_hi, nc) (fit_arc noise, dynspec.py:1097-1101)
 * out: DEVICE double[1]; workspace: 1032 doubles. */
int32_t scint_block_std(const double* a, int64_t ld, int64_t nc, int64_t r0, int64_t r1,
                        int64_t c_lo, int64_t c_hi, double* out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- plain 2-D complex FFT (forward, numpy sign convention) -------------- */
/* Exposed for tests: out may alias in. */
int32_t scint_fft2_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes /*HOST*/);
int32_t scint_fft2(const scint_c128* in, scint_c128* out, int64_t rows, int64_t cols,
                   void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCINT_HIP_H */
