// fft.hip -- 2-D transforms of the hot path built from the two kernels in fft.hpp:
//   scint_fft2              plain complex 2-D FFT (tests, building block)
//   scint_sspec             Dynspec.calc_sspec core      (dynspec.py:3665-3721)
//   scint_cs                conjugate spectrum of a chunk (ththmod.py:777-787)
//   scint_model_from_recov  ifft2(ifftshift(recov)).real  (ththmod.py:322-324)
//   scint_mean / scint_chisq  small deterministic reductions
#include <math.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "fft.hpp"

namespace scint {

// ------------------------------------------------------------------------------
// twiddle cache
// ------------------------------------------------------------------------------
static std::mutex g_tw_mutex;
static std::map<std::pair<int, int64_t>, cplx*> g_tw_cache;  // (device, n) -> table

const cplx* twiddle_table(int64_t n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("scint: hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto key = std::make_pair(dev, n);
    auto it = g_tw_cache.find(key);
    if (it != g_tw_cache.end()) return it->second;
    std::vector<cplx> host((size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int64_t j = 0; j < n; ++j) {
        // exact octant symmetries keep W^{n/4}, W^{n/2}, ... free of rounding noise
        long double a = two_pi * (long double)j / (long double)n;
        host[(size_t)j] = mk((double)cosl(a), (double)-sinl(a));
    }
    if (n % 4 == 0) {
        host[(size_t)(n / 4)] = mk(0.0, -1.0);
        host[(size_t)(n / 2)] = mk(-1.0, 0.0);
        host[(size_t)(3 * n / 4)] = mk(0.0, 1.0);
    } else if (n % 2 == 0) {
        host[(size_t)(n / 2)] = mk(-1.0, 0.0);
    }
    cplx* d = nullptr;
    if (hipMalloc(&d, sizeof(cplx) * (size_t)n) != hipSuccess) {
        set_error("scint: hipMalloc of twiddle table failed");
        return nullptr;
    }
    if (hipMemcpy(d, host.data(), sizeof(cplx) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("scint: hipMemcpy of twiddle table failed");
        hipFree(d);
        return nullptr;
    }
    g_tw_cache[key] = d;
    return d;
}

// ------------------------------------------------------------------------------
// deterministic reductions
// ------------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;

template <class F>
__global__ void __launch_bounds__(256) reduce_partial_kernel(F f, int64_t n, double* partial) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        acc += f(i);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// out[0] = scale * sum(partial[0..np))   (fixed order)
__global__ void __launch_bounds__(256) reduce_final_kernel(const double* partial, int np,
                                                           double scale, double* out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partial[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = acc * scale;
}

template <class F>
static int32_t launch_reduce(F f, int64_t n, double scale, double* partial /*kRedBlocks*/,
                             double* out, hipStream_t stream) {
    int blocks = (int)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, ceil_div(n, 256 * 4)));
    hipLaunchKernelGGL((reduce_partial_kernel<F>), dim3(blocks), dim3(256), 0, stream, f, n, partial);
    SCINT_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, scale, out);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

struct PlainValue {
    const double* x;
    __device__ inline double operator()(int64_t i) const { return x[i]; }
};

// (dyn - m1) * win_t[col] * win_f[row]        (dynspec.py:3667-3674)
struct WindowedValue {
    const double* dyn; const double* wt; const double* wf; const double* m1; int64_t nt;
    __device__ inline double at(int64_t r, int64_t c) const {
        double v = dyn[r * nt + c] - m1[0];
        if (wt) { v = wt[c] * v; v = wf[r] * v; }
        return v;
    }
    __device__ inline double operator()(int64_t i) const {
        const int64_t r = i / nt;
        return at(r, i - r * nt);
    }
};

// ------------------------------------------------------------------------------
// generic rows-pass functors
// ------------------------------------------------------------------------------
// complex array rows; slot = row
struct RowLoadC {
    const cplx* a; int64_t ld;
    __device__ inline cplx operator()(int64_t slot, int j) const { return a[slot * ld + j]; }
};
struct RowStoreC {
    cplx* a; int64_t ld;
    __device__ inline void operator()(int64_t slot, int k, cplx v) const { a[slot * ld + k] = v; }
};

template <class Inner>
struct SlotIsRow {
    Inner in;
    __device__ inline cplx operator()(int64_t s, int j) const { return in(s, j); }
};

// Decimated long rows (n = n1 * n2): slot = row*n1 + j1, element j2 -> x[row][j1 + n1*j2];
// result y[j1][k2] * W_n^{j1 k2} -> dst[row][j1*n2 + k2].  `Inner` maps (row, j) -> cplx.
template <class Inner>
struct DecimLoad {
    Inner in; int n1;
    __device__ inline cplx operator()(int64_t slot, int j2) const {
        const int64_t r = slot / n1;
        const int j1 = (int)(slot - r * n1);
        return in(r, j1 + n1 * j2);
    }
};
struct DecimStore {
    cplx* dst; int64_t ld; int n1; int n2; const cplx* tw_n;  // W_n, n = n1*n2
    __device__ inline void operator()(int64_t slot, int k2, cplx v) const {
        const int64_t r = slot / n1;
        const int j1 = (int)(slot - r * n1);
        const cplx w = tw_n[(int64_t)j1 * k2];
        dst[r * ld + (int64_t)j1 * n2 + k2] = v * w;
    }
};

// FFT of length n along the contiguous axis of `nrows` rows produced by `in(row, j)`,
// written to dst[row][0..n) (row stride ld).  Handles n up to 2^17 by a decimated split.
template <class Inner>
static int32_t rows_fft(Inner in, int64_t nrows, int64_t n, cplx* dst, int64_t ld,
                        hipStream_t stream) {
    SCINT_REQUIRE(is_pow2(n) && n >= 16, "rows_fft: n must be a power of two >= 16");
    if (n <= 8192) {
        return launch_fft_rows(n, nrows, SlotIsRow<Inner>{in}, RowStoreC{dst, ld}, stream);
    }
    const int64_t n2 = 4096, n1 = n / n2;
    SCINT_REQUIRE(n1 <= 32, "rows_fft: n too large (max 131072)");
    const cplx* tw_n = twiddle_table(n);
    if (!tw_n) return SCINT_E_HIP;
    int32_t rc = launch_fft_rows(n2, nrows * n1, DecimLoad<Inner>{in, (int)n1},
                                 DecimStore{dst, ld, (int)n1, (int)n2, tw_n}, stream);
    if (rc != SCINT_OK) return rc;
    // radix-n1 pass over j1 (stride n2) for every (row, k2): view dst as [nrows][n1][n2]
    // batches can exceed the 65535 grid.z limit: chunk them
    for (int64_t b0 = 0; b0 < nrows; b0 += 32768) {
        const int64_t nb = std::min<int64_t>(32768, nrows - b0);
        ArrayLoad al2{dst + b0 * ld, n2, ld};
        ArrayStore as2{dst + b0 * ld, n2, ld};
        rc = run_cols_fft(n1, n2, nb, al2, al2, as2, as2, stream);
        if (rc != SCINT_OK) return rc;
    }
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// scint_fft2
// ------------------------------------------------------------------------------
struct InnerArray {
    const cplx* a; int64_t ld;
    __device__ inline cplx operator()(int64_t r, int j) const { return a[r * ld + j]; }
};

static int32_t fft2_pow2(const cplx* in, cplx* out, int64_t rows, int64_t cols, cplx* ws,
                         hipStream_t stream) {
    int32_t rc = rows_fft(InnerArray{in, cols}, rows, cols, ws, cols, stream);
    if (rc != SCINT_OK) return rc;
    ArrayLoad al{ws, cols, 0};
    ArrayStore as{ws, cols, 0};
    ArrayStore fin{out, cols, 0};
    return run_cols_fft(rows, cols, 1, al, al, as, fin, stream);
}

}  // namespace scint

using namespace scint;

extern "C" int32_t scint_fft2_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "fft2_workspace_bytes: null output");
    SCINT_REQUIRE(rows >= 1 && cols >= 1, "fft2_workspace_bytes: bad shape");
    *bytes = (size_t)rows * (size_t)cols * sizeof(cplx) + 256;
    return SCINT_OK;
}

extern "C" int32_t scint_fft2(const scint_c128* in, scint_c128* out, int64_t rows, int64_t cols,
                              void* workspace, size_t workspace_bytes, void* stream) {
    SCINT_REQUIRE(in && out && workspace, "fft2: null pointer");
    SCINT_REQUIRE(is_pow2(rows) && is_pow2(cols) && rows >= 2 && cols >= 16,
                  "fft2: shape must be powers of two (rows >= 2, cols >= 16)");
    size_t need = 0;
    scint_fft2_workspace_bytes(rows, cols, &need);
    if (workspace_bytes < need) { set_error("scint: fft2 workspace too small"); return SCINT_E_WORKSPACE; }
    return fft2_pow2((const cplx*)in, (cplx*)out, rows, cols, (cplx*)workspace, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------
// scint_mean
// ------------------------------------------------------------------------------
extern "C" int32_t scint_mean(const double* x, int64_t n, double* mean_out, void* stream_) {
    SCINT_REQUIRE(x && mean_out && n > 0, "mean: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    double* scratch = nullptr;
    SCINT_HIP(hipMalloc(&scratch, sizeof(double) * (kRedBlocks + 1)));
    int32_t rc = launch_reduce(PlainValue{x}, n, 1.0 / (double)n, scratch, scratch + kRedBlocks, stream);
    if (rc == SCINT_OK) {
        hipError_t e = hipMemcpyAsync(mean_out, scratch + kRedBlocks, sizeof(double),
                                      hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = hip_fail(e, "mean copy-back", __FILE__, __LINE__);
    }
    hipFree(scratch);
    return rc;
}

// ------------------------------------------------------------------------------
// scint_sspec
// ------------------------------------------------------------------------------
namespace scint {

// rows-pass input of calc_sspec: the (optionally prewhitened) tapered, twice
// mean-subtracted dynamic spectrum, zero-padded on the right (dynspec.py:3667-3685)
struct SspecInner {
    WindowedValue w; const double* m2; int64_t nf_eff, nt_eff; int prewhite;
    __device__ inline double d(int64_t r, int64_t c) const { return w.at(r, c) - m2[0]; }
    __device__ inline cplx operator()(int64_t r, int j) const {
        if (j >= nt_eff) return mk(0.0, 0.0);
        if (!prewhite) return mk(d(r, j), 0.0);
        // convolve2d([[1,-1],[-1,1]], dyn, 'valid')  (dynspec.py:3681)
        return mk(d(r + 1, j + 1) - d(r + 1, j) - d(r, j + 1) + d(r, j), 0.0);
    }
};

// first column pass: rows >= nvalid of the padded array are zero (or a constant-row FFT)
struct PaddedColLoad {
    const cplx* a; int64_t ld; int64_t nvalid; double col0_fill;
    __device__ inline cplx operator()(int64_t, int64_t r, int64_t c) const {
        if (r < nvalid) return a[r * ld + c];
        return mk(c == 0 ? col0_fill : 0.0, 0.0);
    }
};

// last column pass of calc_sspec: |.|^2, fftshift, keep tdel >= 0, post-darken, dB
// (dynspec.py:3686-3689, 3704-3721)
struct SspecStore {
    double* sec; int64_t R, C; int halve; int prewhite; const double* pd_fd; const double* pd_td;
    __device__ inline void operator()(int64_t, int64_t k1, int64_t c, cplx v) const {
        int64_t orow;
        if (halve) {
            if (k1 >= R / 2) return;
            orow = k1;
        } else {
            orow = (k1 + R / 2) % R;
        }
        const int64_t ocol = (c + C / 2) % C;
        double p = v.x * v.x + v.y * v.y;
        if (prewhite) {
            double pd = pd_fd[ocol] * pd_td[orow];
            if (ocol == C / 2 || orow == 0) pd = 1.0;
            p = p / pd;
        }
        sec[orow * C + ocol] = 10.0 * log10(p);
    }
};

}  // namespace scint

extern "C" int32_t scint_sspec_workspace_bytes(int64_t nf, int64_t nt, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "sspec_workspace_bytes: null output");
    SCINT_REQUIRE(nf >= 2 && nt >= 2, "sspec_workspace_bytes: bad shape");
    const int64_t R = 2 * next_pow2(nf), C = 2 * next_pow2(nt);
    *bytes = (size_t)R * (size_t)C * sizeof(cplx) + sizeof(double) * (kRedBlocks + 8) + 1024;
    return SCINT_OK;
}

extern "C" int32_t scint_sspec(const double* dyn, int64_t nf, int64_t nt, const double* win_t,
                               const double* win_f, int32_t prewhite, int32_t halve,
                               const double* pd_fd, const double* pd_td, double* sec_out,
                               void* workspace, size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(dyn && sec_out && workspace, "sspec: null pointer");
    SCINT_REQUIRE(nf >= 2 && nt >= 2, "sspec: bad shape");
    SCINT_REQUIRE((win_t == nullptr) == (win_f == nullptr), "sspec: give both windows or neither");
    SCINT_REQUIRE(!prewhite || halve, "sspec: cannot apply prewhite to full frame");
    SCINT_REQUIRE(!prewhite || (pd_fd && pd_td), "sspec: prewhite needs post-darkening vectors");
    hipStream_t stream = (hipStream_t)stream_;
    size_t need = 0;
    scint_sspec_workspace_bytes(nf, nt, &need);
    if (workspace_bytes < need) { set_error("scint: sspec workspace too small"); return SCINT_E_WORKSPACE; }
    const int64_t R = 2 * next_pow2(nf), C = 2 * next_pow2(nt);  // dynspec.py:3677-3678
    SCINT_REQUIRE(C >= 16, "sspec: nt too small");
    Carver cv(workspace, workspace_bytes);
    cplx* ws = cv.take<cplx>((size_t)R * (size_t)C);
    double* partial = cv.take<double>(kRedBlocks);
    double* scal = cv.take<double>(8);  // [0] = mean1, [1] = mean2

    int32_t rc = launch_reduce(PlainValue{dyn}, nf * nt, 1.0 / (double)(nf * nt), partial, scal, stream);
    if (rc != SCINT_OK) return rc;
    WindowedValue wv{dyn, win_t, win_f, scal, nt};
    rc = launch_reduce(wv, nf * nt, 1.0 / (double)(nf * nt), partial, scal + 1, stream);
    if (rc != SCINT_OK) return rc;

    const int64_t nf_eff = prewhite ? nf - 1 : nf, nt_eff = prewhite ? nt - 1 : nt;
    SspecInner inner{wv, scal + 1, nf_eff, nt_eff, prewhite};
    rc = rows_fft(inner, nf_eff, C, ws, C, stream);
    if (rc != SCINT_OK) return rc;
    PaddedColLoad first{ws, C, nf_eff, 0.0};
    ArrayLoad mid_ld{ws, C, 0};
    ArrayStore mid_st{ws, C, 0};
    SspecStore last{sec_out, R, C, halve, prewhite, pd_fd, pd_td};
    return run_cols_fft(R, C, 1, first, mid_ld, mid_st, last, stream);
}

// ------------------------------------------------------------------------------
// scint_cs
// ------------------------------------------------------------------------------
namespace scint {

struct CsInner {  // np.pad(dspec, right/bottom, constant)  (ththmod.py:777-782)
    const double* dspec; int64_t nt; double pad;
    __device__ inline cplx operator()(int64_t r, int j) const {
        return mk(j < nt ? dspec[r * nt + j] : pad, 0.0);
    }
};

// fftshift on both axes, zero the masked delay rows, optional abs (ththmod.py:786-787, 801)
struct CsStore {
    cplx* cs; int64_t R, C, mask_lo, mask_hi; int incoherent;
    __device__ inline void operator()(int64_t, int64_t k1, int64_t c, cplx v) const {
        const int64_t orow = (k1 + R / 2) % R, ocol = (c + C / 2) % C;
        if (orow >= mask_lo && orow < mask_hi) v = mk(0.0, 0.0);
        if (incoherent) v = mk(hypot(v.x, v.y), 0.0);
        cs[orow * C + ocol] = v;
    }
};

}  // namespace scint

extern "C" int32_t scint_cs_workspace_bytes(int64_t nf, int64_t nt, int64_t npad, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "cs_workspace_bytes: null output");
    SCINT_REQUIRE(nf >= 1 && nt >= 1 && npad >= 0, "cs_workspace_bytes: bad shape");
    const int64_t R = (npad + 1) * nf, C = (npad + 1) * nt;
    *bytes = (size_t)R * (size_t)C * sizeof(cplx) + 1024;
    return SCINT_OK;
}

extern "C" int32_t scint_cs(const double* dspec, int64_t nf, int64_t nt, int64_t npad,
                            double pad_value, int64_t mask_lo, int64_t mask_hi, int32_t incoherent,
                            scint_c128* cs_out, void* workspace, size_t workspace_bytes,
                            void* stream_) {
    SCINT_REQUIRE(dspec && cs_out && workspace, "cs: null pointer");
    SCINT_REQUIRE(nf >= 1 && nt >= 1 && npad >= 0, "cs: bad shape");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t R = (npad + 1) * nf, C = (npad + 1) * nt;
    SCINT_REQUIRE(is_pow2(R) && is_pow2(C) && R >= 2 && C >= 16,
                  "cs: padded shape must be powers of two (rows >= 2, cols >= 16)");
    size_t need = 0;
    scint_cs_workspace_bytes(nf, nt, npad, &need);
    if (workspace_bytes < need) { set_error("scint: cs workspace too small"); return SCINT_E_WORKSPACE; }
    cplx* ws = (cplx*)workspace;
    int32_t rc = rows_fft(CsInner{dspec, nt, pad_value}, nf, C, ws, C, stream);
    if (rc != SCINT_OK) return rc;
    PaddedColLoad first{ws, C, nf, pad_value * (double)C};
    ArrayLoad mid_ld{ws, C, 0};
    ArrayStore mid_st{ws, C, 0};
    CsStore last{(cplx*)cs_out, R, C, mask_lo, mask_hi, incoherent};
    return run_cols_fft(R, C, 1, first, mid_ld, mid_st, last, stream);
}

// ------------------------------------------------------------------------------
// scint_model_from_recov
// ------------------------------------------------------------------------------
namespace scint {

// conj(ifftshift(recov)): real(ifft2(x)) == real(fft2(conj(x))) / (R C)
struct ModelInner {
    const cplx* recov; int64_t R, C;
    __device__ inline cplx operator()(int64_t r, int j) const {
        const int64_t sr = (r + R / 2) % R, sc = ((int64_t)j + C / 2) % C;
        return conj(recov[sr * C + sc]);
    }
};
struct ModelStore {
    double* model; int64_t C; double scale;
    __device__ inline void operator()(int64_t, int64_t k1, int64_t c, cplx v) const {
        model[k1 * C + c] = v.x * scale;
    }
};

}  // namespace scint

extern "C" int32_t scint_model_workspace_bytes(int64_t ntau, int64_t nfd, size_t* bytes) {
    return scint_fft2_workspace_bytes(ntau, nfd, bytes);
}

extern "C" int32_t scint_model_from_recov(const scint_c128* recov, int64_t ntau, int64_t nfd,
                                          double* model_out, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(recov && model_out && workspace, "model: null pointer");
    SCINT_REQUIRE(is_pow2(ntau) && is_pow2(nfd) && ntau >= 2 && nfd >= 16,
                  "model: shape must be powers of two (rows >= 2, cols >= 16)");
    hipStream_t stream = (hipStream_t)stream_;
    size_t need = 0;
    scint_fft2_workspace_bytes(ntau, nfd, &need);
    if (workspace_bytes < need) { set_error("scint: model workspace too small"); return SCINT_E_WORKSPACE; }
    cplx* ws = (cplx*)workspace;
    int32_t rc = rows_fft(ModelInner{(const cplx*)recov, ntau, nfd}, ntau, nfd, ws, nfd, stream);
    if (rc != SCINT_OK) return rc;
    ArrayLoad al{ws, nfd, 0};
    ArrayStore as{ws, nfd, 0};
    ModelStore fin{model_out, nfd, 1.0 / ((double)ntau * (double)nfd)};
    return run_cols_fft(ntau, nfd, 1, al, al, as, fin, stream);
}

// ------------------------------------------------------------------------------
// scint_chisq
// ------------------------------------------------------------------------------
namespace scint {
struct ChisqValue {
    const double* model; int64_t ldm; const double* dspec; int64_t nt; const uint8_t* mask;
    __device__ inline double operator()(int64_t i) const {
        const int64_t r = i / nt, c = i - r * nt;
        const double d = dspec[i];
        const bool use = mask ? (mask[i] != 0) : isfinite(d);
        if (!use) return 0.0;
        const double e = model[r * ldm + c] - d;
        return e * e;
    }
};
}  // namespace scint

extern "C" int32_t scint_chisq(const double* model, int64_t ld_model, const double* dspec,
                               int64_t nf, int64_t nt, const uint8_t* mask, double noise_n,
                               double* out, void* stream_) {
    SCINT_REQUIRE(model && dspec && out && nf > 0 && nt > 0, "chisq: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    double* partial = nullptr;
    SCINT_HIP(hipMalloc(&partial, sizeof(double) * kRedBlocks));
    int32_t rc = launch_reduce(ChisqValue{model, ld_model, dspec, nt, mask}, nf * nt, 1.0 / noise_n,
                               partial, out, stream);
    hipError_t e = hipStreamSynchronize(stream);
    hipFree(partial);
    if (rc == SCINT_OK && e != hipSuccess) rc = hip_fail(e, "chisq sync", __FILE__, __LINE__);
    return rc;
}
