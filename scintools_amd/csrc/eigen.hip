// eigen.hip -- dominant ('largest algebraic') eigenpair of a user-supplied dense Hermitian
// matrix (scint_eigh_top: the eigsh call of modeler).  The eta sweep itself runs on the
// tile-packed storage in eigen_packed.hip.
//
// Replaces scipy.sparse.linalg.eigsh(thth_red, 1, v0=v0, which="LA") of
// Eval_calc (ththmod.py:396-401) and modeler (ththmod.py:308).  theta-theta has a
// zero diagonal, so its spectrum is +/- mixed (lambda_min ~ -0.9 lambda_max): an
// un-shifted power iteration would chase |lambda|.  We run the Hermitian Lanczos
// three-term recurrence from the same start vector ARPACK is given (the middle row),
// take the top Ritz value of the tridiagonal T_k, and stop on the Ritz residual
// beta_k |s_k| <= tol |theta| -- the same quantity ARPACK tests, driven to ~1e-12
// so that |w| agrees with eigsh to better than 1e-9.
//
// Kernels per Lanczos step j (all batched over jobs in blockIdx.y):
//   lanczos_matvec_kernel   u = A q_j - beta_j q_{j-1}; partial q_j^H u     (HBM bound:
//                           16 N^2 B per job; one wavefront owns 8 rows, q_j in LDS,
//                           lanes stride the row -> 1 KiB coalesced per wave-load,
//                           wave-shuffle reduction of the 8 dot products)
//   lanczos_update_kernel   alpha_j = sum partials; w = u - alpha_j q_j; partial |w|^2
// and every `chunk` steps
//   lanczos_check_kernel    top eigenpair of T_k by 64-lane multisection on the Sturm
//                           count + backward recurrence for the eigenvector.
// All reductions use fixed trees / fixed partial order: results are bit-reproducible.
#include <math.h>

#include <algorithm>
#include <vector>

#include "prof.hpp"
#include "thth.hpp"

namespace scint {

constexpr int kRowsPerWave = 8;
constexpr int kRowsPerBlock = 32;      // 4 waves x 8 rows
constexpr int kXChunk = 8192;          // q_j elements staged in LDS at a time (128 KiB)
constexpr int kVecBlock = 256;

struct LanczosJob {
    const cplx* A;      // [n, ld]
    int64_t ld;
    int32_t n;
    int32_t max_steps;  // min(max_iter, n)
    cplx* W[2];         // ping-pong work vectors [n]
    cplx* Q;            // q vectors: [qslots][n]
    int32_t qslots;     // 2 (ring) or max_steps (kept for the eigenvector)
    int32_t pad0;
    double* alpha;      // [max_steps]
    double* beta;       // [max_steps + 1]; beta[0] = |v0|, beta[j] = |w| after step j-1
    double* apart;      // [ceil(n/32)]   partial q_j^H u per matvec block
    double* npart;      // [ceil(n/256)]  partial |w|^2 per vector block
    double* svec;       // [max_steps]    eigenvector of T_k (when the Ritz vector is wanted)
    double* result;     // [4]: theta, resid, -, -
    int32_t* state;     // [2]: done flag, steps used
    double* eig_out;    // |theta| destination (device), may be null
    int32_t* status_out;
    int32_t* iters_out;
    const cplx* v0;     // start vector, or null = row n/2 of A
    double tol;
};

__device__ inline double sum_partials(const double* p, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += p[i];
    return s;
}

// w0 = v0 (or the middle row of A, Eval_calc ththmod.py:398); partial |w0|^2
__global__ void __launch_bounds__(kVecBlock) lanczos_init_kernel(const LanczosJob* jobs) {
    __shared__ double red[kVecBlock / 64];
    const LanczosJob jb = jobs[blockIdx.y];
    const int n = jb.n;
    const int r = blockIdx.x * kVecBlock + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) { jb.state[0] = 0; jb.state[1] = 0; }
    if (blockIdx.x * kVecBlock >= n) return;
    double p = 0.0;
    if (r < n) {
        const cplx v = jb.v0 ? jb.v0[r] : jb.A[(int64_t)(n / 2) * jb.ld + r];
        jb.W[0][r] = v;
        p = norm2(v);
    }
    p = block_sum(p, red);
    if (threadIdx.x == 0) jb.npart[blockIdx.x] = p;
}

__global__ void __launch_bounds__(256) lanczos_matvec_kernel(const LanczosJob* jobs, int step) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* xs = reinterpret_cast<cplx*>(smem_raw);
    __shared__ double red[4];
    const LanczosJob jb = jobs[blockIdx.y];
    const int n = jb.n;
    const int row_base = blockIdx.x * kRowsPerBlock;
    if (n < 2 || row_base >= n || step >= jb.max_steps || jb.state[0]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const cplx* __restrict__ win = (step & 1) ? jb.W[1] : jb.W[0];
    cplx* __restrict__ wout = (step & 1) ? jb.W[0] : jb.W[1];
    const int nvb = (n + kVecBlock - 1) / kVecBlock;
    const double beta = sqrt(sum_partials(jb.npart, nvb));   // |w| of the previous step
    const double inv = beta > 0.0 ? 1.0 / beta : 0.0;
    const int qs = jb.qslots;
    cplx* __restrict__ qcur = jb.Q + (int64_t)(step % qs) * n;
    const cplx* __restrict__ qprev = jb.Q + (int64_t)((step + qs - 1) % qs) * n;

    const int r0 = row_base + wave * kRowsPerWave;
    const cplx* rowp[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int rr = (r0 + r < n) ? (r0 + r) : (n - 1);
        rowp[r] = jb.A + (int64_t)rr * jb.ld;
    }
    cplx acc[kRowsPerWave];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) acc[r] = mk(0.0, 0.0);

    for (int c0 = 0; c0 < n; c0 += kXChunk) {
        const int cn = min(kXChunk, n - c0);
        __syncthreads();
        for (int c = threadIdx.x; c < cn; c += 256) {
            const cplx v = win[c0 + c];
            xs[c] = mk(v.x * inv, v.y * inv);
        }
        __syncthreads();
        int c = lane;
        for (; c + 64 < cn; c += 128) {
            const cplx x0 = xs[c], x1 = xs[c + 64];
            cplx a0[kRowsPerWave], a1[kRowsPerWave];
#pragma unroll
            for (int r = 0; r < kRowsPerWave; ++r) {
                a0[r] = gload_nt(rowp[r] + c0 + c);
                a1[r] = gload_nt(rowp[r] + c0 + c + 64);
            }
#pragma unroll
            for (int r = 0; r < kRowsPerWave; ++r) acc[r] = acc[r] + a0[r] * x0 + a1[r] * x1;
        }
        for (; c < cn; c += 64) {
            const cplx x0 = xs[c];
#pragma unroll
            for (int r = 0; r < kRowsPerWave; ++r) acc[r] = acc[r] + gload_nt(rowp[r] + c0 + c) * x0;
        }
    }
    // the normalised q_j rows owned by this block (needed by the update and by step j+1)
    double ap = 0.0;
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const cplx s = wave_sum(acc[r]);
        const int row = r0 + r;
        if (lane == 0 && row < n) {
            const cplx wv = win[row];
            const cplx q = mk(wv.x * inv, wv.y * inv);
            cplx u = s;
            if (step > 0) {
                const cplx qp = qprev[row];
                u = mk(u.x - beta * qp.x, u.y - beta * qp.y);
            }
            qcur[row] = q;
            wout[row] = u;
            ap += q.x * u.x + q.y * u.y;  // Re(conj(q) u)
        }
    }
    ap = block_sum(ap, red);
    if (threadIdx.x == 0) {
        jb.apart[blockIdx.x] = ap;
        if (blockIdx.x == 0) jb.beta[step] = beta;
    }
}

__global__ void __launch_bounds__(kVecBlock) lanczos_update_kernel(const LanczosJob* jobs, int step) {
    __shared__ double red[kVecBlock / 64];
    const LanczosJob jb = jobs[blockIdx.y];
    const int n = jb.n;
    if (n < 2 || blockIdx.x * kVecBlock >= n || step >= jb.max_steps || jb.state[0]) return;
    const int nmb = (n + kRowsPerBlock - 1) / kRowsPerBlock;
    const double alpha = sum_partials(jb.apart, nmb);
    cplx* __restrict__ w = (step & 1) ? jb.W[0] : jb.W[1];
    const cplx* __restrict__ q = jb.Q + (int64_t)(step % jb.qslots) * n;
    const int r = blockIdx.x * kVecBlock + threadIdx.x;
    double p = 0.0;
    if (r < n) {
        const cplx u = w[r], qv = q[r];
        const cplx v = mk(u.x - alpha * qv.x, u.y - alpha * qv.y);
        w[r] = v;
        p = norm2(v);
    }
    p = block_sum(p, red);
    if (threadIdx.x == 0) {
        jb.npart[blockIdx.x] = p;
        if (blockIdx.x == 0) jb.alpha[step] = alpha;
    }
}

// Number of eigenvalues of T_k (diag a[0..k), off-diagonal b[1..k)) below x.
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

// One wavefront per job: top Ritz pair of T_k after `k` completed steps.
__global__ void __launch_bounds__(64) lanczos_check_kernel(const LanczosJob* jobs, int k_done,
                                                           int final_pass) {
    const LanczosJob jb = jobs[blockIdx.x];
    if (jb.state[0]) return;
    const int lane = threadIdx.x;
    const int n = jb.n;
    if (n < 2) {
        if (lane == 0) {
            jb.state[0] = 1;
            if (jb.status_out) jb.status_out[0] = SCINT_E_EMPTY;
            if (jb.eig_out) jb.eig_out[0] = nan("");
            if (jb.iters_out) jb.iters_out[0] = 0;
        }
        return;
    }
    const int k = min(k_done, jb.max_steps);
    const double* a = jb.alpha;
    const double* b = jb.beta;  // b[i], i >= 1, couples i-1 and i
    const int nvb = (n + kVecBlock - 1) / kVecBlock;
    const double beta_k = sqrt(sum_partials(jb.npart, nvb));
    // Gershgorin bracket
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
    bool finite = isfinite(lo) && isfinite(hi) && isfinite(beta_k);
    double theta = nan(""), resid = nan("");
    if (finite) {
        const double tiny = fmax(scale, 1e-300) * 1e-300 + 1e-300;
        hi = hi + 1e-15 * fabs(hi) + 1e-300;  // count(hi) == k guaranteed
        // multisection: shrink [lo, hi] around the largest eigenvalue (count(lo) < k, count(hi) == k)
        for (int round = 0; round < 40; ++round) {
            const double w = hi - lo;
            if (!(w > 0.0)) break;
            const double x = lo + w * ((double)(lane + 1) / 65.0);
            const int full = (x > lo && x < hi) ? (sturm_count(a, b, k, x, tiny) == k) : 0;
            const unsigned long long m = __ballot(full);
            double nlo, nhi;
            if (m == 0ull) { nlo = __shfl(x, 63, 64); nhi = hi; }
            else {
                const int first = __ffsll((long long)m) - 1;
                nhi = __shfl(x, first, 64);
                nlo = first > 0 ? __shfl(x, first - 1, 64) : lo;
            }
            if (nlo == lo && nhi == hi) break;
            lo = nlo > lo ? nlo : lo;
            hi = nhi < hi ? nhi : hi;
            if (hi - lo <= 4e-16 * fmax(fabs(lo), fabs(hi))) break;
        }
        theta = 0.5 * (lo + hi);
        // eigenvector of T_k for theta by the backward recurrence (stable for the top pair)
        if (lane == 0) {
            double* s = jb.svec;
            double sk = 1.0, skp1 = 0.0, nrm = 0.0, last = 1.0;
            // s[k-1] = 1; b[i] s[i-1] = (theta - a[i]) s[i] - b[i+1] s[i+1]
            if (s) s[k - 1] = 1.0;
            nrm = 1.0;
            for (int i = k - 1; i >= 1; --i) {
                const double bi = b[i];
                double sm1 = (bi != 0.0) ? ((theta - a[i]) * sk - (i + 1 < k ? b[i + 1] * skp1 : 0.0)) / bi : 0.0;
                if (!isfinite(sm1)) sm1 = 0.0;
                if (fabs(sm1) > 1e150) {  // rescale to avoid overflow
                    const double f = 1e-150;
                    sm1 *= f; sk *= f; last *= f; nrm *= f * f;
                    if (s) for (int t = i; t < k; ++t) s[t] *= f;
                }
                if (s) s[i - 1] = sm1;
                nrm += sm1 * sm1;
                skp1 = sk;
                sk = sm1;
            }
            const double inv = 1.0 / sqrt(nrm);
            if (s) for (int t = 0; t < k; ++t) s[t] *= inv;
            resid = beta_k * fabs(last) * inv;
        }
        resid = __shfl(resid, 0, 64);
    }
    if (lane == 0) {
        const bool conv = finite && (resid <= jb.tol * fmax(fabs(theta), 1e-300) || k >= n || beta_k == 0.0);
        const bool stop = conv || !finite || k >= jb.max_steps || final_pass;
        jb.result[0] = theta;
        jb.result[1] = resid;
        if (stop) {
            jb.state[0] = 1;
            jb.state[1] = k;
            if (jb.eig_out) jb.eig_out[0] = fabs(theta);
            if (jb.iters_out) jb.iters_out[0] = k;
            if (jb.status_out)
                jb.status_out[0] = !finite || !isfinite(theta) ? SCINT_E_NONFINITE
                                                                : (conv ? SCINT_OK : SCINT_E_NOCONV);
        }
    }
}

// Ritz vector y = sum_j s_j q_j (un-normalised) + partial |y|^2
__global__ void __launch_bounds__(kVecBlock) lanczos_ritz_kernel(const LanczosJob* jobs, cplx* out) {
    __shared__ double red[kVecBlock / 64];
    const LanczosJob jb = jobs[blockIdx.y];
    const int n = jb.n, k = jb.state[1];
    if (blockIdx.x * kVecBlock >= n) return;
    const int r = blockIdx.x * kVecBlock + threadIdx.x;
    double p = 0.0;
    if (r < n) {
        cplx y = mk(0.0, 0.0);
        for (int j = 0; j < k; ++j) {
            const cplx q = jb.Q[(int64_t)j * n + r];
            const double s = jb.svec[j];
            y = mk(y.x + s * q.x, y.y + s * q.y);
        }
        out[r] = y;
        p = norm2(y);
    }
    p = block_sum(p, red);
    if (threadIdx.x == 0) jb.npart[blockIdx.x] = p;
}

__global__ void __launch_bounds__(kVecBlock) lanczos_scale_kernel(const LanczosJob* jobs, cplx* out,
                                                                  double* w_out) {
    const LanczosJob jb = jobs[blockIdx.y];
    const int n = jb.n;
    const int r = blockIdx.x * kVecBlock + threadIdx.x;
    const int nvb = (n + kVecBlock - 1) / kVecBlock;
    const double nrm = sqrt(sum_partials(jb.npart, nvb));
    const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
    if (r < n) out[r] = mk(out[r].x * inv, out[r].y * inv);
    if (r == 0 && w_out) w_out[0] = jb.result[0];
}

// ------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------
struct JobLayout {   // byte offsets inside one job's slab
    size_t W0, W1, Q, alpha, beta, apart, npart, svec, result, state, total;
};

static JobLayout job_layout(int64_t nmax, int max_steps, int qslots, bool with_A, size_t* a_off) {
    JobLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    if (a_off) *a_off = with_A ? take(sizeof(cplx) * (size_t)nmax * (size_t)nmax) : 0;
    L.W0 = take(sizeof(cplx) * nmax);
    L.W1 = take(sizeof(cplx) * nmax);
    L.Q = take(sizeof(cplx) * (size_t)nmax * (size_t)qslots);
    L.alpha = take(sizeof(double) * (max_steps + 1));
    L.beta = take(sizeof(double) * (max_steps + 2));
    L.apart = take(sizeof(double) * (size_t)ceil_div(nmax, kRowsPerBlock));
    L.npart = take(sizeof(double) * (size_t)ceil_div(nmax, kVecBlock));
    L.svec = take(sizeof(double) * (max_steps + 1));
    L.result = take(sizeof(double) * 4);
    L.state = take(sizeof(int32_t) * 4);
    L.total = align_up(off, 256);
    return L;
}

// Runs Lanczos on `njobs` jobs already uploaded to jobs_dev.  Synchronises the stream
// every `chunk` steps to read the done flags (4 bytes per job).
// Job s keeps its state words at states_dev[4*s .. 4*s+3].
static int32_t run_lanczos(const LanczosJob* jobs_dev, const int32_t* states_dev, int njobs, int nmax,
                           int max_iter, int32_t* flags_pinned, hipStream_t stream) {
    if (njobs == 0) return SCINT_OK;
    const dim3 vgrid((unsigned)ceil_div(nmax, kVecBlock), (unsigned)njobs);
    const dim3 mgrid((unsigned)ceil_div(nmax, kRowsPerBlock), (unsigned)njobs);
    const size_t lds = sizeof(cplx) * (size_t)std::min<int64_t>(nmax, kXChunk);
    if (lds > 64 * 1024)
        SCINT_HIP(hipFuncSetAttribute((const void*)lanczos_matvec_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(lanczos_init_kernel, vgrid, dim3(kVecBlock), 0, stream, jobs_dev);
    SCINT_LAUNCH_CHECK();
    const int steps_cap = std::min(max_iter, nmax);
    int step = 0;
    int chunk = 16;  // first look after 16 steps, then every 8
    while (step < steps_cap) {
        const int upto = std::min(steps_cap, step + chunk);
        for (; step < upto; ++step) {
            const int slot = profiler().begin(kProfMatvec, stream);
            hipLaunchKernelGGL(lanczos_matvec_kernel, mgrid, dim3(256), lds, stream, jobs_dev, step);
            profiler().end(kProfMatvec, slot, stream);
            hipLaunchKernelGGL(lanczos_update_kernel, vgrid, dim3(kVecBlock), 0, stream, jobs_dev, step);
        }
        SCINT_LAUNCH_CHECK();
        hipLaunchKernelGGL(lanczos_check_kernel, dim3((unsigned)njobs), dim3(64), 0, stream, jobs_dev,
                           step, step >= steps_cap ? 1 : 0);
        SCINT_LAUNCH_CHECK();
        SCINT_HIP(hipMemcpyAsync(flags_pinned, states_dev, sizeof(int32_t) * 4 * (size_t)njobs,
                                 hipMemcpyDeviceToHost, stream));
        SCINT_HIP(hipStreamSynchronize(stream));
        if (profiler().enabled) profiler().collect();
        bool all = true;
        for (int i = 0; i < njobs; ++i) all = all && flags_pinned[4 * i] != 0;
        if (all) break;
        chunk = 8;
    }
    return SCINT_OK;
}

}  // namespace scint

using namespace scint;

// ------------------------------------------------------------------------------
// scint_eigh_top
// ------------------------------------------------------------------------------
extern "C" int32_t scint_eigh_top_workspace_bytes(int64_t n, int32_t max_iter, size_t* bytes) {
    SCINT_REQUIRE(bytes && n >= 1 && max_iter >= 1, "eigh_top_workspace_bytes: bad arguments");
    const int steps = (int)std::min<int64_t>(max_iter, n);
    JobLayout L = job_layout(n, steps, steps, false, nullptr);
    *bytes = L.total + sizeof(LanczosJob) + 1024;
    return SCINT_OK;
}

extern "C" int32_t scint_eigh_top(const scint_c128* a, int64_t n, const scint_c128* v0, double tol,
                                  int32_t max_iter, double* w_out, scint_c128* vec_out,
                                  int32_t* status_out, int32_t* iters_out, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(a && w_out && status_out && workspace, "eigh_top: null pointer");
    SCINT_REQUIRE(n >= 1 && max_iter >= 1 && tol > 0, "eigh_top: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    size_t need = 0;
    scint_eigh_top_workspace_bytes(n, max_iter, &need);
    if (workspace_bytes < need) { set_error("scint: eigh_top workspace too small"); return SCINT_E_WORKSPACE; }
    const int steps = (int)std::min<int64_t>(max_iter, n);
    JobLayout L = job_layout(n, steps, steps, false, nullptr);
    char* base = (char*)workspace;
    LanczosJob jb;
    jb.A = (const cplx*)a; jb.ld = n; jb.n = (int32_t)n; jb.max_steps = steps;
    jb.W[0] = (cplx*)(base + L.W0); jb.W[1] = (cplx*)(base + L.W1);
    jb.Q = (cplx*)(base + L.Q); jb.qslots = steps; jb.pad0 = 0;
    jb.alpha = (double*)(base + L.alpha); jb.beta = (double*)(base + L.beta);
    jb.apart = (double*)(base + L.apart); jb.npart = (double*)(base + L.npart);
    jb.svec = (double*)(base + L.svec); jb.result = (double*)(base + L.result);
    jb.state = (int32_t*)(base + L.state);
    jb.eig_out = nullptr; jb.status_out = status_out; jb.iters_out = iters_out;
    jb.v0 = (const cplx*)v0; jb.tol = tol;
    LanczosJob* jd = (LanczosJob*)(base + align_up(L.total, 256));
    SCINT_HIP(hipMemcpyAsync(jd, &jb, sizeof(jb), hipMemcpyHostToDevice, stream));
    int32_t* flags = nullptr;
    SCINT_HIP(hipHostMalloc(&flags, sizeof(int32_t) * 4));
    int32_t rc = run_lanczos(jd, jb.state, 1, (int)n, max_iter, flags, stream);
    (void)hipHostFree(flags);
    if (rc != SCINT_OK) return rc;
    if (vec_out) {
        const dim3 vgrid((unsigned)ceil_div(n, kVecBlock), 1);
        hipLaunchKernelGGL(lanczos_ritz_kernel, vgrid, dim3(kVecBlock), 0, stream, jd, (cplx*)vec_out);
        hipLaunchKernelGGL(lanczos_scale_kernel, vgrid, dim3(kVecBlock), 0, stream, jd, (cplx*)vec_out, w_out);
        SCINT_LAUNCH_CHECK();
    } else {
        SCINT_HIP(hipMemcpyAsync(w_out, jb.result, sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    SCINT_HIP(hipStreamSynchronize(stream));
    return SCINT_OK;
}

