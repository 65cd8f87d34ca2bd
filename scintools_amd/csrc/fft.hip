// fft.hip -- 2-D transforms of the hot path built from the two kernels in fft.hpp:
//   scint_fft2              plain complex 2-D FFT (tests, building block)
//   scint_sspec             Dynspec.calc_sspec core      (dynspec.py:3665-3721)
//   scint_cs                conjugate spectrum of a chunk (ththmod.py:777-787)
//   scint_model_from_recov  ifft2(ifftshift(recov)).real  (ththmod.py:322-324), complex-to-real
//   scint_mean / scint_chisq  small deterministic reductions
//
// Every transform is "source -> row FFT -> column FFT -> sink":
//   RowSource  produces element (row, j) of the padded real/complex input (this is where
//              mean-subtract, window, prewhiten stencil, zero/constant padding, ifftshift
//              are fused -- the padded array is never materialised);
//   ColSource  feeds the first column pass from the row-FFT result, supplying the rows the
//              row pass never computed (all-zero or constant padding rows);
//   ColSink    consumes the final value at natural frequency (k1, k2): fftshift, |.|^2,
//              post-darkening, 10 log10, delay mask, abs, real-part scaling.
// The descriptors are plain structs with a run-time mode (wave-uniform branches); the hot modes
// (calc_sspec, conjugate spectrum, model) are additionally instantiated with the mode as a template
// argument, which takes the 16x-unrolled switch out of those kernels.
// Real input goes two rows per complex transform; the model step is the matching complex-to-real
// transform (model_from_recov below).
//
// Axis lengths that are not powers of two go through Bluestein's chirp-z identity
//   X[k] = w[k] * IFFT_m( FFT_m(x w) * FFT_m(conj w) )[k],   w[j] = exp(-i pi j^2 / n),
// with m = nextpow2(2n-1); both length-m transforms run on the same two kernels and the
// chirp multiplications are fused into the sources/sinks.
#include <math.h>

#include <map>
#include <mutex>
#include <vector>

#include "fft.hpp"
#include "packed.hpp"
#include "prof.hpp"
#include "sspec.hpp"

namespace scint {

// ------------------------------------------------------------------------------
// cached device tables
// ------------------------------------------------------------------------------
static std::mutex g_tab_mutex;
static std::map<std::pair<int, int64_t>, cplx*> g_tw_cache;  // (device, n) -> W_n table

static cplx* upload(const std::vector<cplx>& host) {
    cplx* d = nullptr;
    if (hipMalloc(&d, sizeof(cplx) * host.size()) != hipSuccess) {
        set_error("scint: hipMalloc of an FFT table failed");
        return nullptr;
    }
    if (hipMemcpy(d, host.data(), sizeof(cplx) * host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("scint: hipMemcpy of an FFT table failed");
        (void)hipFree(d);
        return nullptr;
    }
    return d;
}

static const long double kTwoPiL = 6.283185307179586476925286766559005768L;

const cplx* twiddle_table(int64_t n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("scint: hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    auto key = std::make_pair(dev, n);
    auto it = g_tw_cache.find(key);
    if (it != g_tw_cache.end()) return it->second;
    std::vector<cplx> host((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        const long double a = kTwoPiL * (long double)j / (long double)n;
        host[(size_t)j] = mk((double)cosl(a), (double)-sinl(a));
    }
    if (n % 4 == 0) {  // exact quarter turns
        host[(size_t)(n / 4)] = mk(0.0, -1.0);
        host[(size_t)(n / 2)] = mk(-1.0, 0.0);
        host[(size_t)(3 * n / 4)] = mk(0.0, 1.0);
    } else if (n % 2 == 0) {
        host[(size_t)(n / 2)] = mk(-1.0, 0.0);
    }
    cplx* d = upload(host);
    if (d) g_tw_cache[key] = d;
    return d;
}

// Bluestein tables for length n: w[j] = exp(-i pi j^2/n) (j < n) and B = FFT_m(b) with
// b[l] = conj(w[|l|]) wrapped to length m.  Computed in long double on the host.
struct Chirp {
    int64_t n, m;
    const cplx* w;  // [n]
    const cplx* B;  // [m]
};
static std::map<std::pair<int, int64_t>, Chirp> g_chirp_cache;

static void host_fft_ld(std::vector<long double>& re, std::vector<long double>& im) {
    const size_t m = re.size();
    for (size_t i = 1, j = 0; i < m; ++i) {  // bit reversal
        size_t bit = m >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= m; len <<= 1) {
        std::vector<long double> wr(len / 2), wi(len / 2);
        for (size_t k = 0; k < len / 2; ++k) {
            const long double a = kTwoPiL * (long double)k / (long double)len;
            wr[k] = cosl(a);
            wi[k] = -sinl(a);
        }
        for (size_t i = 0; i < m; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const size_t a = i + k, b = i + k + len / 2;
                const long double xr = re[b] * wr[k] - im[b] * wi[k], xi = re[b] * wi[k] + im[b] * wr[k];
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

static const Chirp* chirp_table(int64_t n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("scint: hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    auto key = std::make_pair(dev, n);
    auto it = g_chirp_cache.find(key);
    if (it != g_chirp_cache.end()) return &it->second;
    const int64_t m = std::max<int64_t>(16, next_pow2(2 * n - 1));
    std::vector<long double> wr((size_t)n), wi((size_t)n);
    std::vector<cplx> w((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        const int64_t q = (j * j) % (2 * n);  // angle pi q / n, reduced exactly in integers
        const long double a = kTwoPiL / 2 * (long double)q / (long double)n;
        wr[(size_t)j] = cosl(a);
        wi[(size_t)j] = -sinl(a);
        w[(size_t)j] = mk((double)wr[(size_t)j], (double)wi[(size_t)j]);
    }
    std::vector<long double> br((size_t)m, 0.0L), bi((size_t)m, 0.0L);
    for (int64_t l = 0; l < n; ++l) {
        br[(size_t)l] = wr[(size_t)l];
        bi[(size_t)l] = -wi[(size_t)l];
        if (l > 0) { br[(size_t)(m - l)] = wr[(size_t)l]; bi[(size_t)(m - l)] = -wi[(size_t)l]; }
    }
    host_fft_ld(br, bi);
    std::vector<cplx> B((size_t)m);
    for (int64_t l = 0; l < m; ++l) B[(size_t)l] = mk((double)br[(size_t)l], (double)bi[(size_t)l]);
    Chirp c;
    c.n = n; c.m = m;
    c.w = upload(w);
    c.B = c.w ? upload(B) : nullptr;
    if (!c.w || !c.B) return nullptr;
    auto res = g_chirp_cache.emplace(key, c);
    return &res.first->second;
}

// ------------------------------------------------------------------------------
// deterministic reductions
// ------------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;

// Stage 1 of the deterministic reductions.  Every thread keeps four independent running sums
// (four loads in flight, combined in a fixed order at the end).
template <class F>
__global__ void __launch_bounds__(256) reduce_partial_kernel(F f, int64_t n, double* partial) {
    __shared__ double red[4];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        a0 += f(i); a1 += f(i + stride); a2 += f(i + 2 * stride); a3 += f(i + 3 * stride);
    }
    for (; i < n; i += stride) a0 += f(i);
    const double acc = block_sum((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// The same over a [nrows, ncols] array for functors that need (row, col): block b takes rows
// b, b + G, ...; threads stride the columns (coalesced, no index division).
template <class F>
__global__ void __launch_bounds__(256) reduce2d_partial_kernel(F f, int64_t nrows, int64_t ncols, double* partial) {
    __shared__ double red[4];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        int64_t c = threadIdx.x;
        for (; c + 768 < ncols; c += 1024) {
            a0 += f.at(r, c); a1 += f.at(r, c + 256); a2 += f.at(r, c + 512); a3 += f.at(r, c + 768);
        }
        for (; c < ncols; c += 256) a0 += f.at(r, c);
    }
    const double acc = block_sum((a0 + a1) + (a2 + a3), red);
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

// Both means calc_sspec needs (dynspec.py:3667-3674) in ONE pass over the dynamic spectrum:
//   m1 = mean(dyn),   m2 = mean(w_f[r] w_t[c] (dyn - m1)) = (S_wd - m1 S_w) / n
// with S_wd = sum w_f w_t dyn and S_w = sum w_f w_t: three fixed-order sums, then the scalars.
__global__ void __launch_bounds__(256) sspec_sums_kernel(const double* __restrict__ dyn, const double* __restrict__ wt,
                                                         const double* __restrict__ wf, int64_t nf, int64_t nt,
                                                         double* partial /*3 x gridDim.x*/) {
    __shared__ double red[4];
    double sd = 0.0, swd = 0.0, sw = 0.0;
    for (int64_t r = blockIdx.x; r < nf; r += gridDim.x) {
        const double fr = wf ? wf[r] : 1.0;
        double sd_r = 0.0, swd_r = 0.0, sw_r = 0.0;
        for (int64_t c = threadIdx.x; c < nt; c += 256) {
            const double d = dyn[r * nt + c], w = wt ? wt[c] : 1.0;
            sd_r += d; swd_r += w * d; sw_r += w;
        }
        sd += sd_r; swd += fr * swd_r; sw += fr * sw_r;
    }
    sd = block_sum(sd, red); swd = block_sum(swd, red); sw = block_sum(sw, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = sd; partial[gridDim.x + blockIdx.x] = swd; partial[2 * gridDim.x + blockIdx.x] = sw;
    }
}
__global__ void __launch_bounds__(256) sspec_means_kernel(const double* partial, int np, double n, double* scal) {
    __shared__ double red[4];
    double sd = 0.0, swd = 0.0, sw = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) { sd += partial[i]; swd += partial[np + i]; sw += partial[2 * np + i]; }
    sd = block_sum(sd, red); swd = block_sum(swd, red); sw = block_sum(sw, red);
    if (threadIdx.x == 0) {
        const double m1 = sd / n;
        scal[0] = m1;
        scal[1] = (swd - m1 * sw) / n;
    }
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
template <class F>
static int32_t launch_reduce2d(F f, int64_t nrows, int64_t ncols, double scale, double* partial /*kRedBlocks*/,
                               double* out, hipStream_t stream) {
    const int blocks = (int)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, nrows));
    hipLaunchKernelGGL((reduce2d_partial_kernel<F>), dim3(blocks), dim3(256), 0, stream, f, nrows, ncols, partial);
    SCINT_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, scale, out);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// Scratch for the two-stage reductions, one small buffer per (device, stream): calls on one
// stream are ordered, calls on different streams (one per host thread) never share partials.
static std::map<std::pair<int, hipStream_t>, double*> g_red_scratch;
static double* reduce_scratch(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("scint: hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    const auto key = std::make_pair(dev, stream);
    auto it = g_red_scratch.find(key);
    if (it != g_red_scratch.end()) return it->second;
    double* d = nullptr;
    if (hipMalloc(&d, sizeof(double) * (kRedBlocks + 8)) != hipSuccess) {
        set_error("scint: hipMalloc of reduction scratch failed");
        return nullptr;
    }
    g_red_scratch[key] = d;
    return d;
}

struct PlainValue {
    const double* x;
    __device__ inline double operator()(int64_t i) const { return x[i]; }
};

// (dyn - m1) * win_t[col] * win_f[row]        (dynspec.py:3667-3674)
struct WindowedValue {
    const double* dyn; const double* wt; const double* wf; const double* m1; int nt;
    __device__ inline double at(int r, int c) const {
        double v = dyn[(int64_t)r * nt + c] - m1[0];
        if (wt) { v = wt[c] * v; v = wf[r] * v; }
        return v;
    }
    __device__ inline double operator()(int64_t i) const {
        const int64_t r = i / nt;
        return at((int)r, (int)(i - r * nt));
    }
};

// ------------------------------------------------------------------------------
// sources and sinks
// ------------------------------------------------------------------------------
// Row / column indices, extents and strides are 32-bit in all of these (every transform here is far
// below 2^31 along either axis); the element offset is widened once, at the access.  The `mode` of
// a source or sink is a run-time field; the hot modes are ALSO instantiated with the mode as a
// template argument (RowSourceM / ColSinkM, chosen by with_source / with_sink on the host), which
// removes the switch -- sixteen unrolled copies of it per thread -- from those kernels.

// (i + n/2) mod n for 0 <= i < n without the division an integer `%` costs per element
__device__ inline int shift_half(int i, int n) {
    const int s = i + n / 2;
    return s >= n ? s - n : s;
}

enum SrcMode { SRC_ARRAY = 0, SRC_SSPEC = 1, SRC_CS = 2, SRC_MODEL = 3, SRC_MULCONJ = 4, SRC_CONJ = 5,
               SRC_ACF_IN = 6, SRC_POWER = 7 };

// Element (row, j) of the row-FFT input, j in [0, fft length).
struct RowSource {
    int mode;
    int n_in;            // logical row length; j >= n_in reads as 0 (Bluestein/zero padding)
    const cplx* chirp;   // optional: multiply element j < n_in by chirp[j]
    // SRC_ARRAY: a[row*ld + j];  SRC_MULCONJ: conj(a[row*ld + j] * b[j])
    const cplx* a; int ld; const cplx* b;
    // SRC_SSPEC (dynspec.py:3667-3685)
    WindowedValue wv; const double* m2; int nt_eff; int prewhite;
    // SRC_CS: np.pad(dspec, right, constant)  (ththmod.py:777-782): reuses wv.dyn / wv.nt
    double pad;
    // SRC_MODEL: conj(ifftshift(recov)) -- real(ifft2(x)) == real(fft2(conj x))/(R C)
    int R, C;

    __device__ inline double sspec_d(int r, int c) const { return wv.at(r, c) - m2[0]; }

    template <int M = -1>
    __device__ inline cplx get(int r, int j) const {
        if (j >= n_in) return mk(0.0, 0.0);
        const int m = M >= 0 ? M : mode;
        cplx v;
        switch (m) {
            case SRC_ARRAY: v = a[(int64_t)r * ld + j]; break;
            case SRC_MULCONJ: v = conj(a[(int64_t)r * ld + j] * b[j]); break;
            case SRC_CONJ: v = conj(a[(int64_t)r * ld + j]); break;
            case SRC_ACF_IN:   // (dyn - mean) zero-padded on the right (dynspec.py:3783-3790)
                v = mk(j < wv.nt ? wv.dyn[(int64_t)r * wv.nt + j] - wv.m1[0] : 0.0, 0.0);
                break;
            case SRC_POWER: {  // |X|^2 (dynspec.py:3791-3792); real, so no conjugation is needed
                const cplx x = a[(int64_t)r * ld + j];
                v = mk(x.x * x.x + x.y * x.y, 0.0);
                break;
            }
            case SRC_SSPEC:
                if (j >= nt_eff) v = mk(0.0, 0.0);
                else if (!prewhite) v = mk(sspec_d(r, j), 0.0);
                else  // convolve2d([[1,-1],[-1,1]], dyn, 'valid')  (dynspec.py:3681)
                    v = mk(sspec_d(r + 1, j + 1) - sspec_d(r + 1, j) - sspec_d(r, j + 1) + sspec_d(r, j), 0.0);
                break;
            case SRC_CS: v = mk(j < wv.nt ? wv.dyn[(int64_t)r * wv.nt + j] : pad, 0.0); break;
            default: {  // SRC_MODEL
                const int sr = shift_half(r, R), sc = shift_half(j, C);
                v = conj(a[(int64_t)sr * C + sc]);
            }
        }
        if (chirp) v = v * chirp[j];
        return v;
    }
    __device__ inline cplx operator()(int r, int j) const { return get<-1>(r, j); }
};
template <int M>
struct RowSourceM {
    RowSource s;
    __device__ inline cplx operator()(int r, int j) const { return s.template get<M>(r, j); }
};
// f(source functor): the hot modes get their own instantiation
template <class F>
static int32_t with_source(const RowSource& s, F&& f) {
    switch (s.mode) {
        case SRC_SSPEC: return f(RowSourceM<SRC_SSPEC>{s});
        case SRC_CS: return f(RowSourceM<SRC_CS>{s});
        case SRC_MODEL: return f(RowSourceM<SRC_MODEL>{s});
        default: return f(s);
    }
}

// First column pass: value at (row r, col c) of the row-transformed, padded array.
struct ColSource {
    const cplx* a; int ld;      // row-FFT result
    int nvalid;                 // rows computed by the row pass
    double fill0;               // rows >= nvalid hold fill0 at c == 0, 0 elsewhere
    int n_in;                   // logical column length; r >= n_in reads as 0 (Bluestein padding)
    const cplx* row_post_w;     // rows went through Bluestein: value = w[c] * conj(a) * row_post_scale
    double row_post_scale;
    const cplx* chirp;          // column Bluestein pass 1: multiply by chirp[r]
    const cplx* mulconj_b;      // column Bluestein pass 2: conj(a * b[r]) (then nothing else applies)

    __device__ inline cplx operator()(int64_t, int r, int c) const {
        if (mulconj_b) return conj(a[(int64_t)r * ld + c] * mulconj_b[r]);
        if (r >= n_in) return mk(0.0, 0.0);
        cplx v;
        if (r < nvalid) {
            v = a[(int64_t)r * ld + c];
            if (row_post_w) { v = row_post_w[c] * conj(v); v = v * row_post_scale; }
        } else {
            v = mk(c == 0 ? fill0 : 0.0, 0.0);
        }
        if (chirp) v = v * chirp[r];
        return v;
    }
};
// The common case (power-of-two lengths: no Bluestein factors): rows below nvalid from the array,
// the constant-padded rows from fill0.
struct ColSourcePlain {
    const cplx* a; int ld; int nvalid; double fill0;
    __device__ inline cplx operator()(int64_t, int r, int c) const {
        if (r < nvalid) return a[(int64_t)r * ld + c];
        return mk(c == 0 ? fill0 : 0.0, 0.0);
    }
};

enum SinkMode { SINK_ARRAY = 0, SINK_SSPEC = 1, SINK_CS = 2, SINK_MODEL = 3, SINK_CONJ = 4, SINK_GS_FWD = 5,
                SINK_GS_INV = 6, SINK_ACF = 7 };

// Final value at natural frequency (k1 along the strided axis, c along the contiguous one).
struct ColSink {
    int mode;
    int R, C;                   // logical transform shape
    const cplx* col_post_w;     // column Bluestein: v = w[k1] * conj(v) * col_post_scale, k1 < R only
    double col_post_scale;
    cplx* out_c; int ld;        // SINK_ARRAY / SINK_CS
    double* out_d;              // SINK_SSPEC / SINK_MODEL
    int halve, prewhite; const double* pd_fd; const double* pd_td;   // SINK_SSPEC
    int mask_lo, mask_hi; int incoherent;                            // SINK_CS
    double scale;                                                    // SINK_MODEL / SINK_CONJ / SINK_GS_INV
    int crop_r, crop_c;                                              // SINK_CONJ: keep [0,crop_r) x [0,crop_c)
    int zero_lo, zero_hi;                                            // SINK_GS_FWD: natural rows to zero
    const double* amp; const uint8_t* pos;                           // SINK_GS_INV: sqrt(dyn), posdspec

    int half;                   // real input: only columns 0..C/2 are transformed; the rest is
                                // X[(R-k1)%R, C-c] = conj X[k1, c]

    template <int M = -1>
    __device__ inline void put(int k1, int c, cplx v) const {
        if (k1 >= R) return;
        if (col_post_w) { v = col_post_w[k1] * conj(v); v = v * col_post_scale; }
        if constexpr (M == SINK_SSPEC) {
            // |.|^2 is the same for a value and its conjugate-symmetric partner: one log10 serves both
            // positions (with `halve`, rows k1 and R - k1 never both survive except k1 = 0).
            const double pw = v.x * v.x + v.y * v.y;
            const bool mirror = half && c > 0 && c < C / 2;
            int row[2] = {k1, k1 == 0 ? 0 : R - k1};
            const int col[2] = {shift_half(c, C), shift_half(C - c, C)};
            bool keep[2] = {true, mirror};
            if (halve) {
                keep[0] = row[0] < R / 2;
                keep[1] = mirror && row[1] < R / 2;
            } else {
                row[0] = shift_half(row[0], R);
                row[1] = shift_half(row[1], R);
            }
            // one copy of the (long) log10 sequence per element: the two positions share a loop that
            // is not unrolled; with `halve` at most one of them survives (both only on row 0)
#pragma clang loop unroll(disable)
            for (int e = 0; e < 2; ++e) {
                if (!keep[e]) continue;
                double val = pw;
                if (prewhite) {
                    double pd = pd_fd[col[e]] * pd_td[row[e]];
                    if (col[e] == C / 2 || row[e] == 0) pd = 1.0;
                    val = pw / pd;
                }
                out_d[(int64_t)row[e] * C + col[e]] = 10.0 * log10(val);
            }
            return;
        }
        emit<M>(k1, c, v);
        if (half && c > 0 && c < C / 2) emit<M>(k1 == 0 ? 0 : R - k1, C - c, conj(v));
    }
    __device__ inline void operator()(int64_t, int k1, int c, cplx v) const { put<-1>(k1, c, v); }

    template <int M>
    __device__ inline void emit(int k1, int c, cplx v) const {
        const int m = M >= 0 ? M : mode;
        switch (m) {
            case SINK_ARRAY: out_c[(int64_t)k1 * ld + c] = v; break;
            case SINK_SSPEC: {
                // |.|^2, fftshift, keep tdel >= 0, post-darken, dB (dynspec.py:3686-3721)
                int orow;
                if (halve) {
                    if (k1 >= R / 2) return;
                    orow = k1;
                } else {
                    orow = shift_half(k1, R);
                }
                const int ocol = shift_half(c, C);
                double p = v.x * v.x + v.y * v.y;
                if (prewhite) {
                    double pd = pd_fd[ocol] * pd_td[orow];
                    if (ocol == C / 2 || orow == 0) pd = 1.0;
                    p = p / pd;
                }
                out_d[(int64_t)orow * C + ocol] = 10.0 * log10(p);
                break;
            }
            case SINK_CS: {
                // fftshift on both axes, zero the masked delay rows, optional abs
                // (ththmod.py:786-787, 801)
                const int orow = shift_half(k1, R), ocol = shift_half(c, C);
                if (orow >= mask_lo && orow < mask_hi) v = mk(0.0, 0.0);
                if (incoherent) v = mk(hypot(v.x, v.y), 0.0);
                out_c[(int64_t)orow * C + ocol] = v;
                break;
            }
            case SINK_MODEL:
                if (crop_r > 0) {   // only the [crop_r, crop_c] corner, rows ld apart (chi^2 sweep)
                    if (k1 < crop_r && c < crop_c) out_d[(int64_t)k1 * ld + c] = v.x * scale;
                } else {
                    out_d[(int64_t)k1 * C + c] = v.x * scale;
                }
                break;
            case SINK_CONJ:   // complex inverse transform: ifft2(x) = conj(fft2(conj x)) / (R C), cropped
                if (k1 < crop_r && c < crop_c) out_c[(int64_t)k1 * ld + c] = mk(v.x * scale, -v.y * scale);
                break;
            case SINK_ACF: {   // real(fftshift(ifft2(.)))  (dynspec.py:3793-3795)
                const int orow = shift_half(k1, R), ocol = shift_half(c, C);
                out_d[(int64_t)orow * C + ocol] = v.x * scale;
                break;
            }
            case SINK_GS_FWD:  // CWF[tau < 0] = 0 in natural frequency order (dynspec.py:1869-1870)
                if (k1 >= zero_lo && k1 < zero_hi) v = mk(0.0, 0.0);
                out_c[(int64_t)k1 * ld + c] = v;
                break;
            default: {         // SINK_GS_INV: inverse transform + amplitude constraint (dynspec.py:1871-1875)
                cplx wv = mk(v.x * scale, -v.y * scale);
                const int64_t o = (int64_t)k1 * ld + c;
                if (pos[o]) {
                    // sqrt(dyn) * exp(1j * angle(w)); angle(0) = 0 in NumPy
                    const double mg = hypot(wv.x, wv.y);
                    wv = (mg > 0.0) ? mk(amp[o] * (wv.x / mg), amp[o] * (wv.y / mg)) : mk(amp[o], 0.0);
                }
                out_c[o] = wv;
            }
        }
    }
};
template <int M>
struct ColSinkM {
    ColSink s;
    __device__ inline void operator()(int64_t, int k1, int c, cplx v) const { s.template put<M>(k1, c, v); }
};
template <class F>
static int32_t with_sink(const ColSink& s, F&& f) {
    switch (s.mode) {
        case SINK_SSPEC: return f(ColSinkM<SINK_SSPEC>{s});
        case SINK_CS: return f(ColSinkM<SINK_CS>{s});
        default: return f(s);
    }
}

// ------------------------------------------------------------------------------
// row pass (contiguous axis): power-of-two lengths
// ------------------------------------------------------------------------------
// Loader / Storer concept of fft_rows_kernel: open(slot) -> per-slot accessor; the accessor maps an
// element index of the slot's transform to the value (loader) or stores it (storer).
template <class Src>
struct SlotIsRow {
    Src in;
    struct Slot {
        const SlotIsRow& p; int r;
        __device__ inline cplx operator()(int j) const { return p.in(r, j); }
    };
    __device__ inline Slot open(int64_t s) const { return Slot{*this, (int)s}; }
};
struct RowStoreC {
    static constexpr bool kPair = false;
    cplx* a; int ld;
    struct Slot {
        cplx* row;
        __device__ inline void operator()(int k, cplx v) const { row[k] = v; }
    };
    __device__ inline Slot open(int64_t s) const { return Slot{a + s * ld}; }
};
// Real input, two rows per slot: slot s carries rows 2s and 2s+1 as z = x_{2s} + i x_{2s+1}.
template <class Src>
struct PairLoad {
    Src in; int nrows;
    struct Slot {
        const PairLoad& p; int r0;
        __device__ inline cplx operator()(int j) const {
            return mk(p.in(r0, j).x, r0 + 1 < p.nrows ? p.in(r0 + 1, j).x : 0.0);
        }
    };
    __device__ inline Slot open(int64_t s) const { return Slot{*this, 2 * (int)s}; }
};
// X1[k] = (Z[k] + conj Z[n-k]) / 2,  X2[k] = (Z[k] - conj Z[n-k]) / (2i);  half-width rows
struct PairStore {
    static constexpr bool kPair = true;
    cplx* a; int ld; int nrows;
    struct Slot {
        cplx* row0; cplx* row1;   // row1 == nullptr: the last, unpaired row
        __device__ inline void operator()(int, cplx) const {}
        __device__ inline void pair(int k, cplx zk, cplx zm) const {
            const cplx zc = conj(zm);
            const cplx x1 = mk(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
            const cplx d = mk(zk.x - zc.x, zk.y - zc.y);
            const cplx x2 = mk(0.5 * d.y, -0.5 * d.x);   // d / (2i)
            row0[k] = x1;
            if (row1) row1[k] = x2;
        }
    };
    __device__ inline Slot open(int64_t s) const {
        cplx* r0 = a + 2 * s * ld;
        return Slot{r0, 2 * s + 1 < nrows ? r0 + ld : nullptr};
    }
};
// Decimated long rows (n = n1 * n2): slot = row*n1 + j1, element j2 -> x[row][j1 + n1*j2];
// result y[j1][k2] * W_n^{j1 k2} -> dst[row][j1*n2 + k2].
template <class Src>
struct DecimLoad {
    Src in; int n1;
    struct Slot {
        const DecimLoad& p; int r, j1;
        __device__ inline cplx operator()(int j2) const { return p.in(r, j1 + p.n1 * j2); }
    };
    __device__ inline Slot open(int64_t slot) const {
        const int r = (int)((uint32_t)slot / (uint32_t)n1);   // slots < 2^32
        return Slot{*this, r, (int)slot - r * n1};
    }
};
struct DecimStore {
    static constexpr bool kPair = false;
    cplx* dst; int ld; int n1; int n2; const cplx* tw_n;  // W_n, n = n1*n2
    struct Slot {
        cplx* out; const cplx* tw; int j1;
        __device__ inline void operator()(int k2, cplx v) const { out[k2] = v * tw[j1 * k2]; }
    };
    __device__ inline Slot open(int64_t slot) const {
        const int r = (int)((uint32_t)slot / (uint32_t)n1);
        const int j1 = (int)slot - r * n1;
        return Slot{dst + ((int64_t)r * ld + j1 * n2), tw_n, j1};
    }
};

// FFT of power-of-two length n of `nrows` rows produced by the source functor `src` (value at
// (row, j)), into dst[row][0..n) (row stride ld).  dst must not alias what `src` reads when n > 8192.
template <class Src>
static int32_t rows_fft_src(Src src, int64_t nrows, int64_t n, cplx* dst, int64_t ld, hipStream_t stream) {
    SCINT_REQUIRE(is_pow2(n) && n >= 16, "rows_fft: n must be a power of two >= 16");
    SCINT_REQUIRE(nrows < (1 << 30) && ld < (1 << 30), "rows_fft: extent beyond the 32-bit index range");
    if (n <= 8192) return launch_fft_rows(n, nrows, SlotIsRow<Src>{src}, RowStoreC{dst, (int)ld}, stream);
    const int64_t n2 = 4096, n1 = n / n2;
    SCINT_REQUIRE(n1 <= 32, "rows_fft: n too large (max 131072)");
    SCINT_REQUIRE(nrows * n1 < ((int64_t)1 << 32), "rows_fft: too many decimated rows");
    const cplx* tw_n = twiddle_table(n);
    if (!tw_n) return SCINT_E_HIP;
    int32_t rc = launch_fft_rows<12, 12>(n2, nrows * n1, DecimLoad<Src>{src, (int)n1},
                                         DecimStore{dst, (int)ld, (int)n1, (int)n2, tw_n}, stream);
    if (rc != SCINT_OK) return rc;
    // radix-n1 pass over j1 (stride n2) for every (row, k2): view dst as [nrows][n1][n2];
    // grid.z is limited to 65535, so chunk the rows
    for (int64_t b0 = 0; b0 < nrows; b0 += 32768) {
        const int64_t nb = std::min<int64_t>(32768, nrows - b0);
        ArrayLoad al{dst + b0 * ld, (int)n2, ld};
        ArrayStore as{dst + b0 * ld, (int)n2, ld};
        rc = run_cols_fft(n1, n2, nb, al, al, as, as, stream);
        if (rc != SCINT_OK) return rc;
    }
    return SCINT_OK;
}
static int32_t rows_fft_pow2(const RowSource& in, int64_t nrows, int64_t n, cplx* dst, int64_t ld,
                             hipStream_t stream) {
    return with_source(in, [&](auto src) { return rows_fft_src(src, nrows, n, dst, ld, stream); });
}

// Real rows longer than the in-LDS limit, two at a time: z[s] = x[2s] + i x[2s+1] goes through the
// (decimated) row transform at full width; the first strided pass separates
//   X_{2s}[c] = (Z[c] + conj Z[C-c]) / 2,   X_{2s+1}[c] = (Z[c] - conj Z[C-c]) / (2i),   c <= C/2
// while it loads (two reads per value: the row-transform result is read exactly once in all).
template <class Src>
struct PairRows {
    Src in; int nrows;
    __device__ inline cplx operator()(int s, int j) const {
        const int r0 = 2 * s;
        return mk(in(r0, j).x, r0 + 1 < nrows ? in(r0 + 1, j).x : 0.0);
    }
};
struct PairSplitSource {
    const cplx* z; int ld; int C; int nvalid; double fill0;
    __device__ inline cplx operator()(int64_t, int r, int c) const {
        if (r >= nvalid) return mk(c == 0 ? fill0 : 0.0, 0.0);
        const cplx* row = z + (int64_t)(r >> 1) * ld;
        const cplx zk = row[c], zc = conj(row[c == 0 ? 0 : C - c]);
        return (r & 1) ? mk(0.5 * (zk.y - zc.y), -0.5 * (zk.x - zc.x))
                       : mk(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
    }
};

// ------------------------------------------------------------------------------
// generic 2-D driver
// ------------------------------------------------------------------------------
struct Fft2Plan {
    int64_t R, C;        // logical shape
    int64_t nvalid;      // rows the row pass must transform (the rest: fill0 at col 0)
    int64_t mR, mC;      // power-of-two working lengths (== R/C when those are powers of two)
    bool blue_r, blue_c;
    size_t off_rowA, off_rowB, off_colA, off_colB, total;   // workspace carve (bytes)
};

static Fft2Plan make_plan(int64_t R, int64_t C, int64_t nvalid) {
    Fft2Plan p;
    p.R = R; p.C = C; p.nvalid = nvalid;
    p.blue_r = !is_pow2(R);
    p.blue_c = !(is_pow2(C) && C >= 16);
    p.mR = p.blue_r ? std::max<int64_t>(16, next_pow2(2 * R - 1)) : R;
    p.mC = p.blue_c ? std::max<int64_t>(16, next_pow2(2 * C - 1)) : C;
    size_t off = 0;
    auto take = [&](size_t elems) { off = align_up(off, 256); size_t o = off; off += elems * sizeof(cplx); return o; };
    // row result [nvalid][mC]; Bluestein rows need a second buffer of the same size
    p.off_rowA = take((size_t)nvalid * (size_t)p.mC);
    p.off_rowB = p.blue_c ? take((size_t)nvalid * (size_t)p.mC) : p.off_rowA;
    // column working array [mR][C]; Bluestein columns need a second one
    p.off_colA = take((size_t)p.mR * (size_t)C);
    p.off_colB = p.blue_r ? take((size_t)p.mR * (size_t)C) : p.off_colA;
    p.total = align_up(off, 256);
    return p;
}

// src: RowSource with mode/payload set (n_in, chirp are managed here).
// fill0: value at column 0 of the row-FFT for rows >= nvalid (constant-padded rows), else 0.
// sink: ColSink with mode/payload set (R, C, col_post_*, half are managed here).
// real_input: the source is real -> real-to-complex path when both lengths are powers of two:
//   two rows per row-FFT, only columns 0..C/2 go through the column passes, and the sink emits
//   each value together with its conjugate-symmetric partner.
static int32_t fft2_general(RowSource src, int64_t nvalid, double fill0, int64_t R, int64_t C,
                            ColSink sink, void* workspace, size_t workspace_bytes, hipStream_t stream,
                            bool real_input = false) {
    SCINT_REQUIRE(R >= 1 && C >= 1 && nvalid >= 0 && nvalid <= R, "fft2: bad shape");
    SCINT_REQUIRE(R < (1 << 24) && C < (1 << 24), "fft2: axis longer than 2^24");
    const Fft2Plan p = make_plan(R, C, nvalid);
    if (workspace_bytes < p.total) { set_error("scint: fft workspace too small"); return SCINT_E_WORKSPACE; }
    char* base = (char*)workspace;
    cplx* rowA = (cplx*)(base + p.off_rowA);
    cplx* rowB = (cplx*)(base + p.off_rowB);
    cplx* colA = (cplx*)(base + p.off_colA);
    cplx* colB = (cplx*)(base + p.off_colB);
    int32_t rc;
    sink.half = 0;
    if (real_input && !p.blue_r && !p.blue_c && C <= 8192 && R >= 2) {
        const int64_t Ch = C / 2 + 1;
        // row stride of the half-width arrays: a multiple of 8 elements (128 B), so that the 16-column
        // tiles of the column passes cover whole cache lines (C/2 + 1 is odd); both fit their buffers
        // (rowA holds nvalid x C, colA holds R x C elements)
        const int ChL = (int)((Ch + 7) & ~(int64_t)7);
        src.n_in = (int)C;
        src.chirp = nullptr;
        rc = with_source(src, [&](auto s) {
            return launch_fft_rows(C, (nvalid + 1) / 2, PairLoad<decltype(s)>{s, (int)nvalid},
                                   PairStore{rowA, ChL, (int)nvalid}, stream);
        });
        if (rc != SCINT_OK) return rc;
        ColSourcePlain hs{rowA, ChL, (int)nvalid, fill0};
        sink.R = (int)R; sink.C = (int)C; sink.col_post_w = nullptr; sink.col_post_scale = 1.0; sink.half = 1;
        ArrayLoad mid_ld{colA, ChL, 0};
        ArrayStore mid_st{colA, ChL, 0};
        return with_sink(sink, [&](auto sk) { return run_cols_fft(R, Ch, 1, hs, mid_ld, mid_st, sk, stream); });
    }

    if (real_input && !p.blue_r && !p.blue_c && C > 8192 && R >= 2) {
        // (for rows that fit the LDS the in-kernel separation above is faster: sspec 4096^2 0.72 vs 0.80 ms)
        // long real rows (C = 16384 .. 131072): pairs of rows through the decimated row transform
        const int64_t Ch = C / 2 + 1;
        const int ChL = (int)((Ch + 7) & ~(int64_t)7);
        src.n_in = (int)C;
        src.chirp = nullptr;
        rc = with_source(src, [&](auto s) {
            return rows_fft_src(PairRows<decltype(s)>{s, (int)nvalid}, (nvalid + 1) / 2, C, rowA, C, stream);
        });
        if (rc != SCINT_OK) return rc;
        PairSplitSource hs{rowA, (int)C, (int)C, (int)nvalid, fill0};
        sink.R = (int)R; sink.C = (int)C; sink.col_post_w = nullptr; sink.col_post_scale = 1.0; sink.half = 1;
        ArrayLoad mid_ld{colA, ChL, 0};
        ArrayStore mid_st{colA, ChL, 0};
        return with_sink(sink, [&](auto sk) { return run_cols_fft(R, Ch, 1, hs, mid_ld, mid_st, sk, stream); });
    }

    // ---- rows ---------------------------------------------------------------------
    ColSource cs;
    cs.nvalid = (int)nvalid; cs.fill0 = fill0; cs.n_in = (int)R;
    cs.row_post_w = nullptr; cs.row_post_scale = 1.0; cs.chirp = nullptr; cs.mulconj_b = nullptr;
    src.n_in = (int)C;
    src.chirp = nullptr;
    if (!p.blue_c) {
        rc = rows_fft_pow2(src, nvalid, C, rowA, C, stream);
        if (rc != SCINT_OK) return rc;
        cs.a = rowA; cs.ld = (int)C;
    } else {
        const Chirp* ch = chirp_table(C);
        if (!ch) return SCINT_E_HIP;
        src.chirp = ch->w;
        rc = rows_fft_pow2(src, nvalid, p.mC, rowA, p.mC, stream);
        if (rc != SCINT_OK) return rc;
        RowSource s2;
        s2.mode = SRC_MULCONJ; s2.n_in = (int)p.mC; s2.chirp = nullptr;
        s2.a = rowA; s2.ld = (int)p.mC; s2.b = ch->B;
        rc = rows_fft_pow2(s2, nvalid, p.mC, rowB, p.mC, stream);
        if (rc != SCINT_OK) return rc;
        cs.a = rowB; cs.ld = (int)p.mC;
        cs.row_post_w = ch->w; cs.row_post_scale = 1.0 / (double)p.mC;
    }

    // ---- columns ------------------------------------------------------------------
    sink.R = (int)R; sink.C = (int)C;
    sink.col_post_w = nullptr; sink.col_post_scale = 1.0;
    if (R == 1) {  // nothing to transform along the strided axis
        // a length-1 "FFT": push the row result through the sink with a trivial pass
        SCINT_REQUIRE(false, "fft2: a single-row transform is not supported");
    }
    if (!p.blue_r) {
        ArrayLoad mid_ld{colA, (int)C, 0};
        ArrayStore mid_st{colA, (int)C, 0};
        if (!p.blue_c) {   // no Bluestein factor anywhere: the plain first loader
            ColSourcePlain ps{cs.a, cs.ld, cs.nvalid, cs.fill0};
            return with_sink(sink, [&](auto sk) { return run_cols_fft(R, C, 1, ps, mid_ld, mid_st, sk, stream); });
        }
        return run_cols_fft(R, C, 1, cs, mid_ld, mid_st, sink, stream);
    }
    const Chirp* ch = chirp_table(R);
    if (!ch) return SCINT_E_HIP;
    cs.chirp = ch->w;
    {   // pass 1: FFT_mR(x w) -> colB (mids in place in colA)
        ArrayLoad mid_ld{colA, (int)C, 0};
        ArrayStore mid_st{colA, (int)C, 0};
        ArrayStore to_b{colB, (int)C, 0};
        rc = run_cols_fft(p.mR, C, 1, cs, mid_ld, mid_st, to_b, stream);
        if (rc != SCINT_OK) return rc;
    }
    {   // pass 2: FFT_mR(conj(A B)) -> w[k] conj(.)/mR -> sink  (mids in place in colB... the
        // first pass of run_cols_fft reads colB and writes colA, later passes stay in colA)
        ColSource c2;
        c2.a = colB; c2.ld = (int)C; c2.nvalid = (int)p.mR; c2.fill0 = 0.0; c2.n_in = (int)p.mR;
        c2.row_post_w = nullptr; c2.row_post_scale = 1.0; c2.chirp = nullptr; c2.mulconj_b = ch->B;
        ArrayLoad mid_ld{colA, (int)C, 0};
        ArrayStore mid_st{colA, (int)C, 0};
        sink.col_post_w = ch->w; sink.col_post_scale = 1.0 / (double)p.mR;
        return run_cols_fft(p.mR, C, 1, c2, mid_ld, mid_st, sink, stream);
    }
}

static size_t fft2_general_ws(int64_t R, int64_t C, int64_t nvalid) { return make_plan(R, C, nvalid).total; }

// ------------------------------------------------------------------------------
// model = real(ifft2(ifftshift(recov)))  (ththmod.py:316-317, 357-358) as a complex-to-real transform
// ------------------------------------------------------------------------------
// real(ifft2(X)) = fft2(G) / (R C) with G = conj(Xs), Xs = (X + conj(flip X)) / 2 the Hermitian part
// of X -- an identity for ANY X, so nothing is assumed about the symmetry of the back-mapped
// spectrum.  G is Hermitian, hence fft2(G) is real and half of the work disappears (the mirror image
// of the real-to-complex path above):
//   1. strided-axis FFTs of the columns 0..C/2 of G only (the loader forms G on the fly from the
//      two partner elements of recov: every element of recov is read exactly once);
//   2. every row of the result is Hermitian along the contiguous axis, so two rows a, b go through
//      ONE length-C transform as Z = Y_a + i Y_b; its real part is output row a, its imaginary
//      part output row b.  Only the rows / columns inside the requested corner are produced.
// HBM traffic at 4096^2: 1.2 GB against 2.3 GB for the complex transform.
struct ModelSymSource {
    const cplx* a; int R, C;
    __device__ inline cplx operator()(int64_t, int r, int c) const {
        const int r2 = r == 0 ? 0 : R - r, c2 = c == 0 ? 0 : C - c;
        const cplx x = a[(int64_t)shift_half(r, R) * C + shift_half(c, C)];
        const cplx y = a[(int64_t)shift_half(r2, R) * C + shift_half(c2, C)];
        return mk(0.5 * (x.x + y.x), 0.5 * (y.y - x.y));      // (conj X[r,c] + X[-r,-c]) / 2
    }
};
struct HermPairLoad {
    const cplx* y; int ld; int C;
    struct Slot {
        const cplx* ya; const cplx* yb; int C;
        __device__ inline cplx operator()(int j) const {
            const bool hi = j > C / 2;
            const int jj = hi ? C - j : j;
            cplx a = ya[jj], b = yb[jj];
            if (hi) { a = conj(a); b = conj(b); }
            if (jj == 0 || 2 * jj == C) { a.y = 0.0; b.y = 0.0; }   // real by symmetry: drop the rounding residue
            return mk(a.x - b.y, a.y + b.x);                      // Y_a + i Y_b
        }
    };
    __device__ inline Slot open(int64_t s) const {
        const cplx* r0 = y + 2 * s * ld;
        return Slot{r0, r0 + ld, C};
    }
};
struct RealPairStore {
    static constexpr bool kPair = false;
    double* out; int ld, crop_r, crop_c; double scale;
    struct Slot {
        double* ra; double* rb; int crop_c; double scale;
        __device__ inline void operator()(int k, cplx v) const {
            if (k >= crop_c) return;
            if (ra) ra[k] = v.x * scale;
            if (rb) rb[k] = v.y * scale;
        }
    };
    __device__ inline Slot open(int64_t s) const {
        const int a = 2 * (int)s;
        return Slot{a < crop_r ? out + (int64_t)a * ld : nullptr, a + 1 < crop_r ? out + (int64_t)(a + 1) * ld : nullptr,
                    crop_c, scale};
    }
};

// out[r * ld + c] = real(ifft2(ifftshift(recov)))[r, c] for r < crop_r, c < crop_c; recov is [R][C].
// The same last pass with chi^2 as its sink (ththmod.py:330-368, chi^2 sweep): sum over the cropped corner of
// (model - dspec)^2 where the pixel counts (mask, or finite dspec as ChisqValue below), the model never stored --
// saves its 8 crop_r crop_c bytes out and back in, and a kernel, per curvature.
struct RealPairChisq {
    static constexpr bool kPair = false;
    static constexpr bool kReduce = true;
    const double* dspec; const uint8_t* mask; int ld, crop_r, crop_c; double scale; double* partial;
    struct Slot {
        const double* da; const double* db; const uint8_t* ma; const uint8_t* mb; int crop_c; double scale;
        __device__ static inline double term(double model, double d, const uint8_t* m) {
            const bool use = m ? (*m != 0) : isfinite(d);
            if (!use) return 0.0;
            const double e = model - d;
            return e * e;
        }
        __device__ inline void operator()(int k, cplx v, double& acc) const {
            if (k >= crop_c) return;
            if (da) acc += term(v.x * scale, da[k], ma ? ma + k : nullptr);
            if (db) acc += term(v.y * scale, db[k], mb ? mb + k : nullptr);
        }
    };
    __device__ inline Slot open(int64_t s) const {
        const int a = 2 * (int)s;
        const bool ha = a < crop_r, hb = a + 1 < crop_r;
        return Slot{ha ? dspec + (int64_t)a * ld : nullptr, hb ? dspec + (int64_t)(a + 1) * ld : nullptr,
                    (ha && mask) ? mask + (int64_t)a * ld : nullptr, (hb && mask) ? mask + (int64_t)(a + 1) * ld : nullptr,
                    crop_c, scale};
    }
};

// `fuse` (optional): chi^2 against fuse->dspec instead of storing the model; *fuse_blocks = partial sums written
// to fuse->partial.  Only the power-of-two path fuses (returns with *fuse_blocks = 0 otherwise: model in `out`).
static int32_t model_from_recov(const cplx* recov, int64_t R, int64_t C, double* out, int64_t ld, int64_t crop_r,
                                int64_t crop_c, void* workspace, size_t workspace_bytes, hipStream_t stream,
                                const RealPairChisq* fuse = nullptr, int* fuse_blocks = nullptr) {
    if (fuse_blocks) *fuse_blocks = 0;
    const double scale = 1.0 / ((double)R * (double)C);
    if (!(is_pow2(R) && is_pow2(C) && R >= 2 && C >= 32 && C <= 8192)) {
        // other shapes: the general complex transform (chirp-z where needed), real part at the sink
        RowSource src{};
        src.mode = SRC_MODEL; src.a = recov; src.R = (int)R; src.C = (int)C;
        ColSink sink{};
        sink.mode = SINK_MODEL; sink.out_d = out; sink.scale = scale;
        sink.crop_r = (int)crop_r; sink.crop_c = (int)crop_c; sink.ld = (int)ld;
        return fft2_general(src, R, 0.0, R, C, sink, workspace, workspace_bytes, stream);
    }
    const int64_t Ch = C / 2 + 1;
    const int ChL = (int)((Ch + 7) & ~(int64_t)7);
    // two half-width arrays: the in-place working array of the strided passes and their natural-order
    // result (the last pass permutes rows, so it cannot store in place); both fit the workspace of the
    // general transform (two full-width arrays)
    const size_t half_elems = (size_t)R * (size_t)ChL;
    if (workspace_bytes < 2 * sizeof(cplx) * half_elems) {
        set_error("scint: model workspace too small");
        return SCINT_E_WORKSPACE;
    }
    cplx* work = (cplx*)workspace;
    cplx* half = work + half_elems;
    ArrayLoad mid_ld{work, ChL, 0};
    ArrayStore mid_st{work, ChL, 0};
    ArrayStore last{half, ChL, 0};
    int32_t rc = run_cols_fft(R, Ch, 1, ModelSymSource{recov, (int)R, (int)C}, mid_ld, mid_st, last, stream);
    if (rc != SCINT_OK) return rc;
    const int64_t rows_out = std::min<int64_t>(crop_r, R);
    if (fuse && fuse_blocks) {
        RealPairChisq st = *fuse;
        st.ld = (int)ld; st.crop_r = (int)crop_r; st.crop_c = (int)crop_c; st.scale = scale;
        // one slot per workgroup for C >= 4096 (256 threads per transform), several below: the launcher's rule
        const int tps = (int)(C / kEPT), min_block = C <= 128 ? 128 : 256, block = tps >= min_block ? tps : min_block;
        *fuse_blocks = (int)ceil_div((rows_out + 1) / 2, block / tps);
        return launch_fft_rows<5, 13>(C, (rows_out + 1) / 2, HermPairLoad{half, ChL, (int)C}, st, stream);
    }
    return launch_fft_rows<5, 13>(C, (rows_out + 1) / 2, HermPairLoad{half, ChL, (int)C},
                           RealPairStore{out, (int)ld, (int)crop_r, (int)crop_c, scale}, stream);
}

}  // namespace scint

using namespace scint;

// ------------------------------------------------------------------------------
// scint_fft2
// ------------------------------------------------------------------------------
extern "C" int32_t scint_fft2_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "fft2_workspace_bytes: null output");
    SCINT_REQUIRE(rows >= 2 && cols >= 1, "fft2_workspace_bytes: bad shape");
    *bytes = fft2_general_ws(rows, cols, rows) + 256;
    return SCINT_OK;
}

extern "C" int32_t scint_fft2(const scint_c128* in, scint_c128* out, int64_t rows, int64_t cols,
                              void* workspace, size_t workspace_bytes, void* stream) {
    SCINT_REQUIRE(in && out && workspace, "fft2: null pointer");
    SCINT_REQUIRE(rows >= 2 && cols >= 1, "fft2: bad shape");
    RowSource src{};
    src.mode = SRC_ARRAY; src.a = (const cplx*)in; src.ld = cols;
    ColSink sink{};
    sink.mode = SINK_ARRAY; sink.out_c = (cplx*)out; sink.ld = cols;
    return fft2_general(src, rows, 0.0, rows, cols, sink, workspace, workspace_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------
// scint_mean
// ------------------------------------------------------------------------------
extern "C" int32_t scint_mean(const double* x, int64_t n, double* mean_out, void* stream_) {
    SCINT_REQUIRE(x && mean_out && n > 0, "mean: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    double* scratch = reduce_scratch(stream);
    if (!scratch) return SCINT_E_HIP;
    int32_t rc = launch_reduce(PlainValue{x}, n, 1.0 / (double)n, scratch, scratch + kRedBlocks, stream);
    if (rc == SCINT_OK) {
        hipError_t e = hipMemcpyAsync(mean_out, scratch + kRedBlocks, sizeof(double),
                                      hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = hip_fail(e, "mean copy-back", __FILE__, __LINE__);
    }
    return rc;
}

// ------------------------------------------------------------------------------
// scint_sspec
// ------------------------------------------------------------------------------
extern "C" int32_t scint_sspec_workspace_bytes(int64_t nf, int64_t nt, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "sspec_workspace_bytes: null output");
    SCINT_REQUIRE(nf >= 2 && nt >= 2, "sspec_workspace_bytes: bad shape");
    const int64_t R = 2 * next_pow2(nf), C = std::max<int64_t>(16, 2 * next_pow2(nt));
    // the generic path's buffers also hold the fast path's intermediate (sspec.hip): [R/2][nt] <= [nf][C]
    *bytes = std::max(fft2_general_ws(R, C, nf), sspec_fast_workspace(nf, nt)) + sizeof(double) * (3 * kRedBlocks + 8) + 2048;
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
    SCINT_REQUIRE(C >= 16, "sspec: nt must be at least 5");
    if (sspec_fast_supported(nf, nt, halve))      // two HBM round trips instead of four (sspec.hip)
        return sspec_fast(dyn, nf, nt, win_t, win_f, prewhite, pd_fd, pd_td, sec_out, workspace, stream);
    const size_t fft_bytes = fft2_general_ws(R, C, nf);
    Carver cv((char*)workspace + align_up(fft_bytes, 256), workspace_bytes - align_up(fft_bytes, 256));
    double* partial = cv.take<double>(3 * kRedBlocks);
    double* scal = cv.take<double>(8);  // [0] = mean1, [1] = mean2

    {
        const int blocks = (int)std::min<int64_t>(kRedBlocks, nf);
        hipLaunchKernelGGL(sspec_sums_kernel, dim3(blocks), dim3(256), 0, stream, dyn, win_t, win_f, nf, nt, partial);
        hipLaunchKernelGGL(sspec_means_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, (double)(nf * nt), scal);
        SCINT_LAUNCH_CHECK();
    }
    WindowedValue wv{dyn, win_t, win_f, scal, (int)nt};
    const int64_t nf_eff = prewhite ? nf - 1 : nf, nt_eff = prewhite ? nt - 1 : nt;
    RowSource src{};
    src.mode = SRC_SSPEC; src.wv = wv; src.m2 = scal + 1; src.nt_eff = nt_eff; src.prewhite = prewhite;
    ColSink sink{};
    sink.mode = SINK_SSPEC; sink.out_d = sec_out; sink.halve = halve; sink.prewhite = prewhite;
    sink.pd_fd = pd_fd; sink.pd_td = pd_td;
    return fft2_general(src, nf_eff, 0.0, R, C, sink, workspace, fft_bytes, stream, /*real_input=*/true);
}

// ------------------------------------------------------------------------------
// scint_cs
// ------------------------------------------------------------------------------
extern "C" int32_t scint_cs_workspace_bytes(int64_t nf, int64_t nt, int64_t npad, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "cs_workspace_bytes: null output");
    SCINT_REQUIRE(nf >= 1 && nt >= 1 && npad >= 0, "cs_workspace_bytes: bad shape");
    const int64_t R = (npad + 1) * nf, C = (npad + 1) * nt;
    SCINT_REQUIRE(R >= 2, "cs_workspace_bytes: need at least two delay rows");
    *bytes = fft2_general_ws(R, C, nf) + 1024;
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
    SCINT_REQUIRE(R >= 2, "cs: need at least two delay rows");
    RowSource src{};
    src.mode = SRC_CS; src.wv.dyn = dspec; src.wv.nt = nt; src.pad = pad_value;
    ColSink sink{};
    sink.mode = SINK_CS; sink.out_c = (cplx*)cs_out; sink.ld = C;
    sink.mask_lo = mask_lo; sink.mask_hi = mask_hi; sink.incoherent = incoherent;
    // rows >= nf are constant rows of pad_value: their FFT is C*pad at column 0
    return fft2_general(src, nf, pad_value * (double)C, R, C, sink, workspace, workspace_bytes, stream,
                        /*real_input=*/true);
}

// ------------------------------------------------------------------------------
// scint_model_from_recov
// ------------------------------------------------------------------------------
extern "C" int32_t scint_model_workspace_bytes(int64_t ntau, int64_t nfd, size_t* bytes) {
    return scint_fft2_workspace_bytes(ntau, nfd, bytes);
}

extern "C" int32_t scint_model_from_recov(const scint_c128* recov, int64_t ntau, int64_t nfd,
                                          double* model_out, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(recov && model_out && workspace, "model: null pointer");
    SCINT_REQUIRE(ntau >= 2 && nfd >= 1, "model: bad shape");
    SCINT_REQUIRE(ntau < (1 << 24) && nfd < (1 << 24), "model: axis longer than 2^24");
    return model_from_recov((const cplx*)recov, ntau, nfd, model_out, nfd, ntau, nfd, workspace, workspace_bytes,
                            (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------
// scint_chisq
// ------------------------------------------------------------------------------
namespace scint {
struct ChisqValue {
    const double* model; int64_t ldm; const double* dspec; int64_t nt; const uint8_t* mask;
    __device__ inline double at(int64_t r, int64_t c) const {
        const int64_t i = r * nt + c;
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
    double* partial = reduce_scratch(stream);
    if (!partial) return SCINT_E_HIP;
    return launch_reduce2d(ChisqValue{model, ld_model, dspec, nt, mask}, nf, nt, 1.0 / noise_n, partial, out, stream);
}

// ------------------------------------------------------------------------------
// scint_ifft2_shifted: complex ifft2(ifftshift(x)), optionally cropped and scaled
// (single_chunk_retrieval, ththmod.py:1465-1468)
// ------------------------------------------------------------------------------
extern "C" int32_t scint_ifft2_shifted(const scint_c128* in, int64_t rows, int64_t cols, double scale,
                                       int64_t crop_rows, int64_t crop_cols, scint_c128* out,
                                       void* workspace, size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(in && out && workspace, "ifft2_shifted: null pointer");
    SCINT_REQUIRE(rows >= 2 && cols >= 1 && crop_rows >= 1 && crop_rows <= rows && crop_cols >= 1 && crop_cols <= cols,
                  "ifft2_shifted: bad shape");
    RowSource src{};
    src.mode = SRC_MODEL; src.a = (const cplx*)in; src.R = rows; src.C = cols;   // conj(ifftshift(x))
    ColSink sink{};
    sink.mode = SINK_CONJ; sink.out_c = (cplx*)out; sink.ld = crop_cols;
    sink.scale = scale / ((double)rows * (double)cols);
    sink.crop_r = crop_rows; sink.crop_c = crop_cols;
    return fft2_general(src, rows, 0.0, rows, cols, sink, workspace, workspace_bytes, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------
// scint_gerchberg_saxton: Dynspec.gerchberg_saxton iterations (dynspec.py:1868-1875)
// ------------------------------------------------------------------------------
extern "C" int32_t scint_gs_workspace_bytes(int64_t rows, int64_t cols, size_t* bytes) {
    SCINT_REQUIRE(bytes && rows >= 2 && cols >= 1, "gs_workspace_bytes: bad shape");
    *bytes = fft2_general_ws(rows, cols, rows) + sizeof(cplx) * (size_t)rows * (size_t)cols + 1024;
    return SCINT_OK;
}

extern "C" int32_t scint_gerchberg_saxton(scint_c128* wavefield, int64_t rows, int64_t cols,
                                          const double* amp, const uint8_t* pos, int64_t zero_lo,
                                          int64_t zero_hi, int32_t niter, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(wavefield && amp && pos && workspace, "gerchberg_saxton: null pointer");
    SCINT_REQUIRE(rows >= 2 && cols >= 1 && niter >= 0, "gerchberg_saxton: bad arguments");
    size_t need = 0;
    scint_gs_workspace_bytes(rows, cols, &need);
    if (workspace_bytes < need) { set_error("scint: gerchberg_saxton workspace too small"); return SCINT_E_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t fft_bytes = align_up(fft2_general_ws(rows, cols, rows), 256);
    cplx* cwf = (cplx*)((char*)workspace + fft_bytes);
    for (int it = 0; it < niter; ++it) {
        RowSource f{};
        f.mode = SRC_ARRAY; f.a = (const cplx*)wavefield; f.ld = cols;
        ColSink fs{};
        fs.mode = SINK_GS_FWD; fs.out_c = cwf; fs.ld = cols; fs.zero_lo = zero_lo; fs.zero_hi = zero_hi;
        int32_t rc = fft2_general(f, rows, 0.0, rows, cols, fs, workspace, fft_bytes, stream);
        if (rc != SCINT_OK) return rc;
        RowSource b{};
        b.mode = SRC_CONJ; b.a = cwf; b.ld = cols;
        ColSink bs{};
        bs.mode = SINK_GS_INV; bs.out_c = (cplx*)wavefield; bs.ld = cols;
        bs.scale = 1.0 / ((double)rows * (double)cols); bs.amp = amp; bs.pos = pos;
        rc = fft2_general(b, rows, 0.0, rows, cols, bs, workspace, fft_bytes, stream);
        if (rc != SCINT_OK) return rc;
    }
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// scint_acf: Dynspec.calc_acf(method='direct') (dynspec.py:3780-3797)
// ------------------------------------------------------------------------------
namespace scint {
struct MaxAbsValue {   // arr /= np.max(arr): the maximum of a real array via two-stage max
    const double* x;
};
__global__ void __launch_bounds__(256) max_partial_kernel(const double* x, int64_t n, double* partial) {
    __shared__ double red[4];
    double m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, x[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__global__ void __launch_bounds__(256) scale_by_max_kernel(double* x, int64_t n, const double* partial, int np) {
    double m = -INFINITY;
    for (int i = 0; i < np; ++i) m = fmax(m, partial[i]);
    const double inv = 1.0 / m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = x[i] / m;
    (void)inv;
}
}  // namespace scint

extern "C" int32_t scint_acf_workspace_bytes(int64_t nf, int64_t nt, size_t* bytes) {
    SCINT_REQUIRE(bytes && nf >= 1 && nt >= 1, "acf_workspace_bytes: bad shape");
    const int64_t R = 2 * nf, C = 2 * nt;
    *bytes = fft2_general_ws(R, C, R) + sizeof(cplx) * (size_t)R * (size_t)C + sizeof(double) * (kRedBlocks + 8) + 1024;
    return SCINT_OK;
}

extern "C" int32_t scint_acf(const double* dyn, int64_t nf, int64_t nt, int32_t subtract_mean,
                             int32_t normalise, double* acf_out, void* workspace, size_t workspace_bytes,
                             void* stream_) {
    SCINT_REQUIRE(dyn && acf_out && workspace, "acf: null pointer");
    SCINT_REQUIRE(nf >= 1 && nt >= 1, "acf: bad shape");
    size_t need = 0;
    scint_acf_workspace_bytes(nf, nt, &need);
    if (workspace_bytes < need) { set_error("scint: acf workspace too small"); return SCINT_E_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t R = 2 * nf, C = 2 * nt;
    const size_t fft_bytes = align_up(fft2_general_ws(R, C, R), 256);
    cplx* spec = (cplx*)((char*)workspace + fft_bytes);
    Carver cv((char*)workspace + fft_bytes + sizeof(cplx) * (size_t)R * (size_t)C,
              workspace_bytes - fft_bytes - sizeof(cplx) * (size_t)R * (size_t)C);
    double* partial = cv.take<double>(kRedBlocks);
    double* scal = cv.take<double>(8);
    int32_t rc = SCINT_OK;
    if (subtract_mean) rc = launch_reduce(PlainValue{dyn}, nf * nt, 1.0 / (double)(nf * nt), partial, scal, stream);
    else SCINT_HIP(hipMemsetAsync(scal, 0, sizeof(double), stream));
    if (rc != SCINT_OK) return rc;
    RowSource f{};
    f.mode = SRC_ACF_IN; f.wv.dyn = dyn; f.wv.nt = nt; f.wv.m1 = scal;
    ColSink fs{};
    fs.mode = SINK_ARRAY; fs.out_c = spec; fs.ld = C;
    rc = fft2_general(f, nf, 0.0, R, C, fs, workspace, fft_bytes, stream, /*real_input=*/true);
    if (rc != SCINT_OK) return rc;
    RowSource b{};
    b.mode = SRC_POWER; b.a = spec; b.ld = C;
    ColSink bs{};
    bs.mode = SINK_ACF; bs.out_d = acf_out; bs.scale = 1.0 / ((double)R * (double)C);
    // the power spectrum is real and even, so fft2 and ifft2 of it coincide up to the 1/(RC) factor
    rc = fft2_general(b, R, 0.0, R, C, bs, workspace, fft_bytes, stream, /*real_input=*/true);
    if (rc != SCINT_OK || !normalise) return rc;
    const int blocks = (int)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, ceil_div(R * C, 1024)));
    hipLaunchKernelGGL(max_partial_kernel, dim3(blocks), dim3(256), 0, stream, acf_out, R * C, partial);
    hipLaunchKernelGGL(scale_by_max_kernel, dim3(blocks), dim3(256), 0, stream, acf_out, R * C, partial, blocks);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// scint_chisq_sweep: chisq_calc(modeler(...)) for every curvature in ONE call
// (the loop [chisq_calc(dspec, CS, tau, fd, eta, edges, N, mask) for eta in etas], ththmod.py:330-368)
// ------------------------------------------------------------------------------
namespace scint {

// out[c][r] = in[r][c] through a 32 x 33 LDS tile (coalesced on both sides)
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, int64_t rows, int64_t cols, T* __restrict__ out) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = in[(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) out[(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// The model step of one retired curvature, chained on the sweep's tail stream while the Lanczos
// steps of the resident curvatures continue: rank-1 back-map of |w| V V^H -> inverse FFT (model
// dynamic spectrum, [nf, nt] corner only) -> chi^2 reduction into chisq_out[e].
// Everything between the back-map and the reduction is kept TRANSPOSED (recov^T, model^T against
// dspec^T): ifft2 commutes with the transpose, and a back-map workgroup owns one Doppler column
// of recov, which is one contiguous row of recov^T -- coalesced stores instead of 16-B stores
// 16 nfd bytes apart.
constexpr int kChisqPartials = 8192;   // per-workgroup chi^2 sums of the fused model step (one per two output rows, at most)

// chi^2 WITHOUT the model transform (round 4).  When the conjugate spectrum has the shape of the dynamic spectrum
// (npad = 0: model = real(ifft2(ifftshift(recov))) is not cropped) and every pixel counts (no mask, dspec finite),
// Parseval's identity gives
//     sum_{f,t} (model - dspec)^2 = 1/(R C) sum_k | Xs[k] - D[k] |^2,
//     Xs[k] = (X[k] + conj X[-k]) / 2  = fft2(model)   (X = ifftshift(recov): the real part of ifft2 is the transform
//                                                        of the Hermitian part, for ANY recov),
//     D = fft2(dspec)                                   (formed once per sweep from the dynamic spectrum itself),
// so a curvature's chi^2 is one pass over recov and D instead of a 2-D inverse transform (1.2 GB of traffic at 4096^2)
// plus a reduction: both sides are compared in fftshift-ed coordinates, where recov lives and where the library's own
// conjugate-spectrum routine leaves D.  Everything is kept transposed as in the tail below: A = recov^T [P = nfd][Q = ntau],
// S = fftshift(fft2(dspec^T)) = fftshift(fft2(dspec))^T.  The partner of (p, q) is (2 hp - p mod P, 2 hq - q mod Q),
// hp = P / 2, hq = Q / 2 (integer halves: the origin of the shifted axes); dspec is real, so the terms of (p, q) and
// of its partner are equal: a workgroup takes a row p of the half hp <= p < P (and row 0 when P is even) and counts it
// twice unless the row is its own partner.  Every element of recov is read once (as itself or as a partner), D on
// the half plane only: 0.40 GB at 4096^2.  Fixed summation order: bit-reproducible.
// Round 5: (a) several curvatures per launch (grid.y; RevBatch of thth.hpp -- the per-curvature tail cost the sweep's host thread 11 API
// calls per curvature), (b) only the delay rows q of the curvature's band are read: recov is exactly 0 outside it (and not even
// written there by the batched back-map), so those rows contribute sum |D|^2, which comes from prefix / suffix sums of
//     colsum[q] = sum_{p in the half plane} weight_p |S[p][q]|^2      (chisq_colsum_kernel, chisq_prefix_kernel: once per sweep).
// The band is symmetric about q = Q / 2 and does not contain row 0 unless it is the whole axis (rev_prep_batch_kernel), so
// the partner row of a row in the band is in the band.
__global__ void __launch_bounds__(256) chisq_parseval_batch_kernel(RevBatch bt, const RevJobDev* __restrict__ jobs,
                                                                   const cplx* __restrict__ S, int P, int Q,
                                                                   const double* __restrict__ pre, const double* __restrict__ suf,
                                                                   double* __restrict__ partial_base, int partial_stride) {
    __shared__ double red[4];
    const cplx* __restrict__ A = bt.recov[blockIdx.y];
    const unsigned long long* __restrict__ bound = jobs[bt.job[blockIdx.y]].bound;
    double* __restrict__ partial = partial_base + (int64_t)blockIdx.y * partial_stride;
    const int qlo = (int)bound[kRevBandLo], qhi = (int)bound[kRevBandHi];
    const int hp = P / 2, hq = Q / 2, nhalf = hp + 1;
    double tot = 0.0;
    for (int hb = (int)blockIdx.x; hb < nhalf; hb += (int)gridDim.x) {
        const int p = hp + hb < P ? hp + hb : 0;
        int pp = 2 * hp - p; pp += pp < 0 ? P : 0; pp -= pp >= P ? P : 0;
        const cplx* __restrict__ a = A + (int64_t)p * Q;
        const cplx* __restrict__ b = A + (int64_t)pp * Q;
        const cplx* __restrict__ d = S + (int64_t)p * Q;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        auto term = [&](int q) {
            int qq = 2 * hq - q; qq += qq < 0 ? Q : 0; qq -= qq >= Q ? Q : 0;
            const cplx x = gload(a + q), y = gload(b + qq), z = gload(d + q);
            const double re = 0.5 * (x.x + y.x) - z.x, im = 0.5 * (x.y - y.y) - z.y;
            return re * re + im * im;
        };
        int q = qlo + (int)threadIdx.x;
        for (; q + 768 <= qhi; q += 1024) { a0 += term(q); a1 += term(q + 256); a2 += term(q + 512); a3 += term(q + 768); }
        for (; q <= qhi; q += 256) a0 += term(q);
        const double row = (a0 + a1) + (a2 + a3);
        tot += pp == p ? row : 2.0 * row;
    }
    tot = block_sum(tot, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = tot;
        if (blockIdx.x == 0) {                         // the rows outside the band: |D|^2 alone
            partial[gridDim.x] = gload(pre + qlo);
            partial[gridDim.x + 1] = gload(suf + qhi + 1);
        }
    }
}
// colsum[q] of the kernel above: threads along q (coalesced); the half-plane rows in kColsumChunks contiguous chunks
// (grid.y), each summed in the kernel's own order into part[chunk][q]; chisq_prefix_kernel adds the chunks in order
constexpr int kColsumChunks = 64;
__global__ void __launch_bounds__(256) chisq_colsum_kernel(const cplx* __restrict__ S, int P, int Q, double* __restrict__ part) {
    const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (q >= Q) return;
    const int hp = P / 2, nhalf = hp + 1;
    const int per = (nhalf + kColsumChunks - 1) / kColsumChunks, h0 = (int)blockIdx.y * per, h1 = min(h0 + per, nhalf);
    double acc = 0.0;
    for (int hb = h0; hb < h1; ++hb) {
        const int p = hp + hb < P ? hp + hb : 0;
        int pp = 2 * hp - p; pp += pp < 0 ? P : 0; pp -= pp >= P ? P : 0;
        const cplx z = gload(S + (int64_t)p * Q + q);
        const double v = z.x * z.x + z.y * z.y;
        acc += pp == p ? v : 2.0 * v;
    }
    part[(int64_t)blockIdx.y * Q + q] = acc;
}
// ---- chi^2 from the back-map's accumulators (round 6; thth.hpp: RevFuse) --------------------------------------------------------
// On axes that are symmetric about 0 (even lengths, x0 = -(n / 2) step: every fft_axis) the rank-1 Hermitian back-map is
// mirror-symmetric -- pair (j, i) falls in pixel (P - p, Q - q) when (i, j) falls in (p, q), with the conjugate weight -- unless a
// pair sits ON a bin edge (the back-map raises the curvature's flag then).  So for every INTERIOR pixel (p >= 1, q >= 1) the
// Hermitian part Xs[k] of the identity above is recov itself (to its rounding: the two sums have different orders), and the
// pixel's chi^2 term |recov - D|^2 is formed by the back-map workgroup that holds it in LDS: the image is neither written nor read
// back (0.46 of 0.55 GB per curvature at the headline size).  What is left for this side:
//   * the EDGE set -- Doppler column p = 0 and delay row q = 0, whose mirror pixels are not mirror images (their partners'
//     pairs fall off the axes): the back-map still writes them, chisq_edge_batch_kernel forms their terms with the partner
//     formula; its partners (0, Q - q) and (P - p, 0) are in the edge set;
//   * the interior rows outside the curvature's band: sum |D|^2 from prefix / suffix sums of
//     colsumI[q] = sum_{p >= 1} |S[p][q]|^2 (q >= 1; 0 at q = 0), formed once per sweep by the two kernels below.
__global__ void __launch_bounds__(256) chisq_colsum_interior_kernel(const cplx* __restrict__ S, int P, int Q, double* __restrict__ part) {
    const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (q >= Q) return;
    const int per = (P - 1 + kColsumChunks - 1) / kColsumChunks, p0 = 1 + (int)blockIdx.y * per, p1 = min(p0 + per, P);
    double acc = 0.0;
    for (int p = p0; p < p1; ++p) {
        const cplx z = gload(S + (int64_t)p * Q + q);
        acc += z.x * z.x + z.y * z.y;
    }
    part[(int64_t)blockIdx.y * Q + q] = q == 0 ? 0.0 : acc;
}
// partial[image][items .. items + 2] = the edge set's terms, the interior rows below the band, the interior rows above it
__global__ void __launch_bounds__(256) chisq_edge_batch_kernel(RevBatch bt, const RevJobDev* __restrict__ jobs, const cplx* __restrict__ S,
                                                               int P, int Q, const double* __restrict__ preI, const double* __restrict__ sufI,
                                                               double* __restrict__ partial_base, int64_t partial_stride, int items) {
    __shared__ double red[4];
    const cplx* __restrict__ A = bt.recov[blockIdx.x];
    const unsigned long long* __restrict__ bound = jobs[bt.job[blockIdx.x]].bound;
    const int qlo = (int)bound[kRevBandLo], qhi = (int)bound[kRevBandHi];
    auto at = [&](int p, int q) {                       // recov^T, 0 outside the band (not written there)
        return (q >= qlo && q <= qhi) ? gload(A + (int64_t)p * Q + q) : mk(0.0, 0.0);
    };
    auto term = [&](int p, int q) {
        const int pp = p == 0 ? 0 : P - p, qq = q == 0 ? 0 : Q - q;
        const cplx x = at(p, q), y = at(pp, qq), z = gload(S + (int64_t)p * Q + q);
        const double re = 0.5 * (x.x + y.x) - z.x, im = 0.5 * (x.y - y.y) - z.y;
        return re * re + im * im;
    };
    double acc = 0.0;
    for (int q = (int)threadIdx.x; q < Q; q += 256) acc += term(0, q);
    for (int p = 1 + (int)threadIdx.x; p < P; p += 256) acc += term(p, 0);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        double* partial = partial_base + (int64_t)blockIdx.x * partial_stride + items;
        partial[0] = acc;
        partial[1] = gload(preI + qlo);
        partial[2] = gload(sufI + qhi + 1);
    }
}

// pre[q] = sum_{q' < q} colsum[q'], suf[q] = sum_{q' >= q} colsum[q'], q = 0 .. Q: one workgroup, a contiguous segment per
// thread, fixed order
__global__ void __launch_bounds__(256) chisq_prefix_kernel(const double* __restrict__ part, double* __restrict__ colsum, int Q,
                                                           double* __restrict__ pre, double* __restrict__ suf) {
    __shared__ double seg[256], base_pre[256], base_suf[256];
    const int t = (int)threadIdx.x, len = (Q + 255) / 256, q0 = min(t * len, Q), q1 = min(q0 + len, Q);
    double acc = 0.0;
    for (int q = q0; q < q1; ++q) {
        double c = 0.0;
        for (int k = 0; k < kColsumChunks; ++k) c += part[(int64_t)k * Q + q];
        colsum[q] = c;
        acc += c;
    }
    seg[t] = acc;
    __syncthreads();
    if (t == 0) {
        double run = 0.0;
        for (int k = 0; k < 256; ++k) { base_pre[k] = run; run += seg[k]; }
        run = 0.0;
        for (int k = 255; k >= 0; --k) { base_suf[k] = run; run += seg[k]; }      // sum of the segments AFTER k
    }
    __syncthreads();
    double run = base_pre[t];
    for (int q = q0; q < q1; ++q) { pre[q] = run; run += colsum[q]; }
    run = base_suf[t];
    for (int q = q1 - 1; q >= q0; --q) { run += colsum[q]; suf[q] = run; }
    if (t == 255) { pre[Q] = base_pre[255] + seg[255]; suf[Q] = 0.0; }
}
// chisq_out[job] = scale * sum(partial[0 .. np)) for every curvature of the batch (fixed order)
__global__ void __launch_bounds__(256) chisq_final_batch_kernel(RevBatch bt, const double* __restrict__ partial_base, int partial_stride,
                                                                int np, double scale, double* __restrict__ chisq_out) {
    __shared__ double red[4];
    const double* __restrict__ partial = partial_base + (int64_t)blockIdx.x * partial_stride;
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) chisq_out[bt.job[blockIdx.x]] = acc * scale;
}
// 1.0 per non-finite value (the mask chisq_calc builds when none is given, ththmod.py:357-358, is then not all-true)
struct NonFiniteValue {
    const double* x;
    __device__ double operator()(int64_t i) const { const double v = x[i]; return (v - v == 0.0) ? 0.0 : 1.0; }
};

struct ChisqTail : SweepTail {
    GeomDev g; const int32_t* keep_n; const double* etas;
    const cplx* vec; int64_t vstride; const double* w; const double* th_red; int64_t M;
    const double* dspecT; int64_t nf, nt; const uint8_t* maskT; double noise_n; double* chisq_out;
    const cplx* specT = nullptr;      // fftshift(fft2(dspec^T)): set when chi^2 goes by Parseval (no model transform)
    const RevJobDev* jobs_dev = nullptr; const double* pre = nullptr; const double* suf = nullptr;   // Parseval route: per-curvature table, |D|^2 sums
    std::vector<uint8_t> uniform;     // per curvature: its back-map takes the uniform-grid kernel (launch_rev_uniform's flags)
    bool fuse = false;                // ... and chi^2 comes from its accumulators (RevFuse: symmetric axes)
    const double* preI = nullptr; const double* sufI = nullptr; int32_t* asym = nullptr; int64_t partial_stride = 0;
    int images = 1;                   // image buffers (and partial-sum sets) per tail lane
    // one set of scratch buffers per tail lane
    cplx* recovT_[kTailLanes]; double* modelT_[kTailLanes]; void* fft_ws_[kTailLanes]; size_t fft_ws_bytes;
    double* partial_[kTailLanes]; void* rev_scratch_[kTailLanes];

    int batch_max() const override { return specT ? images : 1; }

    // Parseval route: the curvatures a chunk retired, <= `images` of them, in four launches (thth.hpp)
    int32_t retire_batch(const int64_t* es, int count, hipStream_t tail, int lane) override {
        if (!specT) return SweepTail::retire_batch(es, count, tail, lane);
        RevBatch b;
        b.n = 0; b.pad = 0;
        const size_t image = (size_t)g.ntau * (size_t)g.nfd;
        for (int k = 0; k < count; ++k) {
            if (keep_n[es[k]] < 3) continue;           // crop to fewer than three centres: chi^2 stays NaN (no mean edge step)
            SCINT_REQUIRE(b.n < images && b.n < kRevBatchMax, "chisq sweep: tail batch too large");
            b.job[b.n] = (int32_t)es[k];
            b.recov[b.n] = recovT_[lane] + image * (size_t)b.n;
            ++b.n;
        }
        for (int k = b.n; k < kRevBatchMax; ++k) { b.job[k] = 0; b.recov[k] = nullptr; }
        if (b.n == 0) return SCINT_OK;
        const int ps = profiler().begin(kProfRevmap, tail);
        RevFuse fz{specT, partial_[lane], partial_stride, asym};
        RevBatch general, fused;
        int32_t rc = launch_rev_map_rank1_batch(jobs_dev, b, g, uniform.empty() ? nullptr : uniform.data(), fuse ? &fz : nullptr,
                                                &general, &fused, tail);
        profiler().end(kProfRevmap, ps, tail);
        if (rc != SCINT_OK) return rc;
        const int pm = profiler().begin(kProfModel, tail);
        const int P = (int)g.nfd, Q = (int)g.ntau;
        const double scale = 1.0 / ((double)P * (double)Q) / noise_n;
        if (!fuse) {                                   // every image was written: one Parseval pass over the whole batch
            general = b;
            fused.n = 0;
        }
        if (fused.n > 0) {
            // the uniform-grid curvatures: their interior terms are in partial[image][0 .. items) already
            const int items = (int)rev_diag_items(g);
            hipLaunchKernelGGL(chisq_edge_batch_kernel, dim3((unsigned)fused.n), dim3(256), 0, tail, fused, jobs_dev, specT, P, Q, preI, sufI,
                               partial_[lane], partial_stride, items);
            hipLaunchKernelGGL(chisq_final_batch_kernel, dim3((unsigned)fused.n), dim3(256), 0, tail, fused, partial_[lane], (int)partial_stride,
                               items + 3, scale, chisq_out);
        }
        if (general.n > 0) {
            double* part = partial_[lane] + (int64_t)fused.n * partial_stride;      // (behind the fused images' sums)
            const int nblk = std::min(P / 2 + 1, kChisqPartials);
            hipLaunchKernelGGL(chisq_parseval_batch_kernel, dim3((unsigned)nblk, (unsigned)general.n), dim3(256), 0, tail, general, jobs_dev, specT,
                               P, Q, pre, suf, part, (int)partial_stride);
            hipLaunchKernelGGL(chisq_final_batch_kernel, dim3((unsigned)general.n), dim3(256), 0, tail, general, part, (int)partial_stride,
                               nblk + 2, scale, chisq_out);
        }
        if (hipGetLastError() != hipSuccess) rc = SCINT_E_HIP;
        profiler().end(kProfModel, pm, tail);
        return rc;
    }

    // model route (a cropped model, a mask, a non-finite dspec): one curvature at a time through the complex-to-real transform
    int32_t retire(int64_t e, hipStream_t tail, int lane) override {
        const int64_t n = keep_n[e];
        if (n < 3) return SCINT_OK;                       // crop to fewer than three centres: chi^2 stays NaN
        cplx* recovT = recovT_[lane]; double* modelT = modelT_[lane]; void* fft_ws = fft_ws_[lane];
        double* partial = partial_[lane]; void* rev_scratch = rev_scratch_[lane];
        const int ps = profiler().begin(kProfRevmap, tail);
        int32_t rc = launch_rev_map_rank1(vec + e * vstride, w + e, th_red + e * M, n, g, etas[e], recovT, true,
                                          rev_scratch, tail);
        profiler().end(kProfRevmap, ps, tail);
        if (rc != SCINT_OK) return rc;
        const int pm = profiler().begin(kProfModel, tail);
        RealPairChisq fuse{}; fuse.dspec = dspecT; fuse.mask = maskT; fuse.partial = partial;
        int nblk = 0;
        rc = model_from_recov(recovT, g.nfd, g.ntau, modelT, nf, nt, nf, fft_ws, fft_ws_bytes, tail,
                              nt <= 2 * (int64_t)kChisqPartials ? &fuse : nullptr, &nblk);
        if (rc == SCINT_OK && nblk > 0) {     // chi^2 came out of the model transform's last pass: add its per-workgroup sums in order
            hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, tail, partial, nblk, 1.0 / noise_n, chisq_out + e);
            if (hipGetLastError() != hipSuccess) rc = SCINT_E_HIP;
        } else if (rc == SCINT_OK) {
            rc = launch_reduce2d(ChisqValue{modelT, nf, dspecT, nf, maskT}, nt, nf, 1.0 / noise_n, partial, chisq_out + e, tail);
        }
        profiler().end(kProfModel, pm, tail);
        return rc;
    }
};

// The shared partner table trusts the caller's crop_group (a public entry point: include/scint_hip.h): members of a group must
// hold the SAME reduced centres.  Checked on the device before a table is shared: row `rows[b]` of th_red against row `rows[0]`,
// n values, bit for bit; any difference clears the group's flag.
struct RowList { int32_t rows[256]; };
__global__ void __launch_bounds__(256) rows_equal_kernel(const double* __restrict__ th_red, int64_t ld, RowList rl, int n, int32_t* flag) {
    const unsigned long long* a = (const unsigned long long*)(th_red + (int64_t)rl.rows[0] * ld);
    const unsigned long long* b = (const unsigned long long*)(th_red + (int64_t)rl.rows[blockIdx.x] * ld);
    bool diff = false;
    for (int k = threadIdx.x; k < n; k += 256) diff |= a[k] != b[k];
    if (diff) atomicExch(flag, 0);
}
constexpr int kRevWalkTables = 2;      // crops that get a partner table (the largest same-crop groups of a sweep)
constexpr int kRevWalkMinGroup = 8;    // ... if at least this many curvatures share the crop (a table costs about one back-map's walk)
struct ChisqSweepLayout {
    size_t recov[kTailLanes], model[kTailLanes], dspecT, maskT, specT, fft[kTailLanes], partial[kTailLanes], rev[kTailLanes], sweep, total, fft_bytes, sweep_bytes;
    size_t jobs, bounds, colsum, pre, suf;     // Parseval route: RevJobDev table, per-curvature constants, |D|^2 sums along the delay axis
    size_t uflags;                             // ... and the grid test's flag of every curvature (thth.hpp: launch_rev_uniform)
    size_t colsumI, preI, sufI, asym;          // chi^2 from the back-map's accumulators (RevFuse): |D|^2 sums of the interior, per-curvature flags
    int64_t partial_stride;                    // doubles per image of a lane's partial sums
    size_t walk[kRevWalkTables];               // partner tables of the back-map for the largest groups of same-crop curvatures (thth.hpp)
    int images;                                // image buffers per tail lane (tail batches)
};
// Image buffers per tail lane: as many curvatures as a tail batch may hold (thth.hpp), within 8 GiB over all lanes (8 per lane at
// 4096^2, 4 at 8192^2)
static int chisq_images_per_lane(int64_t ntau, int64_t nfd, int64_t nf, int64_t nt) {
    if (!(nf == ntau && nt == nfd)) return 1;                      // cropped model: the per-curvature model route only
    const size_t image = sizeof(cplx) * (size_t)ntau * (size_t)nfd;
    return (int)std::max<size_t>(1, std::min<size_t>(kRevBatchMax, (((size_t)8 << 30) / kTailLanes) / std::max<size_t>(image, 1)));
}
static int32_t chisq_sweep_layout(int64_t M, int64_t neta, int64_t batch, int32_t max_iter, int64_t ntau, int64_t nfd,
                                  int64_t nf, int64_t nt, ChisqSweepLayout* L) {
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); size_t o = off; off += bytes; return o; };
    L->dspecT = take(sizeof(double) * (size_t)nf * (size_t)nt);
    L->maskT = take((size_t)nf * (size_t)nt);
    L->specT = take((nf == ntau && nt == nfd) ? sizeof(cplx) * (size_t)ntau * (size_t)nfd : 0);   // Parseval route only
    L->fft_bytes = fft2_general_ws(nfd, ntau, nfd);
    L->images = chisq_images_per_lane(ntau, nfd, nf, nt);
    const bool parseval_shape = nf == ntau && nt == nfd;
    L->jobs = take(parseval_shape ? sizeof(RevJobDev) * (size_t)neta : 0);
    L->bounds = take(parseval_shape ? sizeof(unsigned long long) * kRevWords * (size_t)neta : 0);
    L->uflags = take(parseval_shape ? sizeof(int32_t) * (size_t)neta : 0);
    L->colsumI = take(parseval_shape ? sizeof(double) * (size_t)(ntau + 1) * (size_t)(kColsumChunks + 1) : 0);
    L->preI = take(parseval_shape ? sizeof(double) * (size_t)(ntau + 1) : 0);
    L->sufI = take(parseval_shape ? sizeof(double) * (size_t)(ntau + 1) : 0);
    L->asym = take(parseval_shape ? sizeof(int32_t) * (size_t)neta : 0);
    L->partial_stride = std::max<int64_t>(kChisqPartials, parseval_shape ? rev_diag_items_for(ntau, nfd) : 0) + 8;
    L->colsum = take(parseval_shape ? sizeof(double) * (size_t)(ntau + 1) * (size_t)(kColsumChunks + 1) : 0);   // [0]: the sums; then the chunks' partials
    L->pre = take(parseval_shape ? sizeof(double) * (size_t)(ntau + 1) : 0);
    L->suf = take(parseval_shape ? sizeof(double) * (size_t)(ntau + 1) : 0);
    for (int t = 0; t < kRevWalkTables; ++t) L->walk[t] = take(parseval_shape ? (size_t)nfd * (size_t)(M + 1) : 0);    // masks [nfd][M], then col_ok [nfd]
    for (int l = 0; l < kTailLanes; ++l) {
        L->recov[l] = take(sizeof(cplx) * (size_t)ntau * (size_t)nfd * (size_t)L->images);
        L->model[l] = take(sizeof(double) * (size_t)nf * (size_t)nt);
        L->fft[l] = take(L->fft_bytes);
        L->partial[l] = take(sizeof(double) * (size_t)L->partial_stride * (size_t)L->images);
        L->rev[l] = take(256);
    }
    int32_t rc = sweep_workspace_bytes(M, neta, batch, max_iter, true, 1, &L->sweep_bytes);
    if (rc != SCINT_OK) return rc;
    L->sweep = take(L->sweep_bytes);
    L->total = align_up(off, 256);
    return SCINT_OK;
}

}  // namespace scint

extern "C" int32_t scint_chisq_sweep_workspace_bytes(int64_t M, int64_t neta, int64_t batch, int32_t max_iter,
                                                     int64_t ntau, int64_t nfd, int64_t nf, int64_t nt, size_t* bytes) {
    SCINT_REQUIRE(bytes && ntau >= 2 && nfd >= 1 && nf >= 1 && nt >= 1 && nf <= ntau && nt <= nfd,
                  "chisq_sweep_workspace_bytes: bad arguments");
    ChisqSweepLayout L;
    const int32_t rc = chisq_sweep_layout(M, neta, batch, max_iter, ntau, nfd, nf, nt, &L);
    if (rc == SCINT_OK) *bytes = L.total + 256;
    return rc;
}

namespace scint { static int32_t g_chisq_fused = 0; static int64_t g_chisq_redone = 0; }
extern "C" int32_t scint_chisq_sweep_last_route(int32_t* fused, int64_t* redone) {
    SCINT_REQUIRE(fused && redone, "chisq_sweep_last_route: null output");
    *fused = g_chisq_fused; *redone = g_chisq_redone;
    return SCINT_OK;
}

extern "C" int32_t scint_chisq_sweep(const scint_c128* cs, const scint_cs_geom* geom, const double* th_cents,
                                     int64_t M, const int32_t* keep_idx, const int32_t* keep_n,
                                     const double* etas, int64_t neta, double tol, int32_t max_iter,
                                     int64_t batch, const double* th_red, const int32_t* crop_group, const double* dspec, int64_t nf,
                                     int64_t nt, const uint8_t* mask, double noise_n, double* chisq_out,
                                     double* w_out, scint_c128* vec_out, int64_t vec_stride,
                                     int32_t* status_out, int32_t* iters_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    SCINT_REQUIRE(cs && geom && th_red && dspec && chisq_out && w_out && vec_out && workspace && keep_n && etas,
                  "chisq_sweep: null pointer");
    SCINT_REQUIRE(nf >= 1 && nt >= 1 && nf <= geom->ntau && nt <= geom->nfd && noise_n != 0.0, "chisq_sweep: bad shape");
    ChisqSweepLayout L;
    int32_t rc = chisq_sweep_layout(M, neta, batch, max_iter, geom->ntau, geom->nfd, nf, nt, &L);
    if (rc != SCINT_OK) return rc;
    if (workspace_bytes < L.total) { set_error("scint: chisq_sweep workspace too small"); return SCINT_E_WORKSPACE; }
    char* base = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    // dspec^T (and mask^T) once per call, before the sweep starts (the tail stream is ordered after
    // the caller's stream by run_sweep)
    double* dspecT = (double*)(base + L.dspecT);
    uint8_t* maskT = mask ? (uint8_t*)(base + L.maskT) : nullptr;
    {
        const dim3 grid((unsigned)ceil_div(nt, 32), (unsigned)ceil_div(nf, 32));
        SCINT_REQUIRE(grid.y <= 65535, "chisq_sweep: nf too large");
        hipLaunchKernelGGL((transpose_kernel<double>), grid, dim3(256), 0, st, dspec, nf, nt, dspecT);
        if (mask) hipLaunchKernelGGL((transpose_kernel<uint8_t>), grid, dim3(256), 0, st, mask, nf, nt, maskT);
        SCINT_LAUNCH_CHECK();
        // a curvature whose crop leaves nothing keeps NaN (all-ones bit pattern), as chisq_calc's nan_to_num-free path does
        SCINT_HIP(hipMemsetAsync(chisq_out, 0xff, sizeof(double) * (size_t)neta, st));
    }
    ChisqTail t;
    t.g = to_dev(*geom); t.keep_n = keep_n; t.etas = etas;
    t.vec = (const cplx*)vec_out; t.vstride = vec_stride; t.w = w_out; t.th_red = th_red; t.M = M;
    t.dspecT = dspecT; t.nf = nf; t.nt = nt; t.maskT = maskT; t.noise_n = noise_n; t.chisq_out = chisq_out;
    t.fft_ws_bytes = L.fft_bytes;
    for (int l = 0; l < kTailLanes; ++l) {
        t.recovT_[l] = (cplx*)(base + L.recov[l]); t.modelT_[l] = (double*)(base + L.model[l]);
        t.fft_ws_[l] = base + L.fft[l];
        t.partial_[l] = (double*)(base + L.partial[l]); t.rev_scratch_[l] = base + L.rev[l];
    }
    if (nf == geom->ntau && nt == geom->nfd && !mask && nf >= 2 && nt >= 2 &&
        geom->ntau <= INT32_MAX / 2 && geom->nfd <= INT32_MAX / 2) {
        // chi^2 by Parseval (chisq_parseval_kernel): possible when every pixel of an uncropped model counts.  One
        // host round trip per sweep decides it (a non-finite pixel of dspec leaves chisq_calc's default mask).
        double* cnt = t.partial_[0] + kChisqPartials;
        rc = launch_reduce(NonFiniteValue{dspec}, nf * nt, 1.0, t.partial_[0], cnt, st);
        if (rc != SCINT_OK) return rc;
        double bad = 1.0;
        SCINT_HIP(hipMemcpyAsync(&bad, cnt, sizeof(double), hipMemcpyDeviceToHost, st));
        SCINT_HIP(hipStreamSynchronize(st));
        if (bad == 0.0) {
            cplx* specT = (cplx*)(base + L.specT);
            rc = scint_cs(dspecT, nt, nf, 0, 0.0, 0, 0, 0, (scint_c128*)specT, t.fft_ws_[0], L.fft_bytes, stream);
            if (rc != SCINT_OK) return rc;
            t.specT = specT;
            // |D|^2 summed over the Doppler half plane for every delay row, and its prefix / suffix sums: what the rows
            // outside a curvature's band contribute to its chi^2
            const int P = (int)geom->nfd, Q = (int)geom->ntau;
            double* colsum = (double*)(base + L.colsum);
            double* part = colsum + (Q + 1);
            hipLaunchKernelGGL(chisq_colsum_kernel, dim3((unsigned)ceil_div(Q, 256), (unsigned)kColsumChunks), dim3(256), 0, st, specT, P, Q, part);
            hipLaunchKernelGGL(chisq_prefix_kernel, dim3(1), dim3(256), 0, st, part, colsum, Q, (double*)(base + L.pre), (double*)(base + L.suf));
            SCINT_LAUNCH_CHECK();
            t.pre = (const double*)(base + L.pre); t.suf = (const double*)(base + L.suf);
            t.partial_stride = L.partial_stride;
            // chi^2 from the back-map's accumulators: axes symmetric about 0 to 1e-7 of a step (np.histogram2d's edges mirror each other
            // that well; the back-map itself settles every pair that is closer to an edge), even lengths.  SCINT_CHISQ_FUSE=0: never.
            {
                const GeomDev& gd = t.g;
                const char* env = getenv("SCINT_CHISQ_FUSE");
                const bool sym = P % 2 == 0 && Q % 2 == 0 && gd.fd1_step > 0.0 && gd.tau1_step > 0.0 &&
                                 fabs((double)P * gd.fd1_step + 2.0 * gd.fd0) <= 1e-7 * gd.fd1_step &&
                                 fabs((double)Q * gd.tau1_step + 2.0 * gd.tau0) <= 1e-7 * gd.tau1_step;
                if (sym && !(env && env[0] == '0')) {
                    double* colsumI = (double*)(base + L.colsumI);
                    double* partI = colsumI + (Q + 1);
                    hipLaunchKernelGGL(chisq_colsum_interior_kernel, dim3((unsigned)ceil_div(Q, 256), (unsigned)kColsumChunks), dim3(256), 0, st, specT, P, Q, partI);
                    hipLaunchKernelGGL(chisq_prefix_kernel, dim3(1), dim3(256), 0, st, partI, colsumI, Q, (double*)(base + L.preI), (double*)(base + L.sufI));
                    SCINT_LAUNCH_CHECK();
                    SCINT_HIP(hipMemsetAsync(base + L.asym, 0, sizeof(int32_t) * (size_t)neta, st));
                    t.preI = (const double*)(base + L.preI); t.sufI = (const double*)(base + L.sufI);
                    t.asym = (int32_t*)(base + L.asym);
                    t.fuse = true;
                }
            }
            // the per-curvature table of the batched tail (thth.hpp): everything but the image buffer is known now
            std::vector<RevJobDev> table((size_t)neta);
            unsigned long long* bounds = (unsigned long long*)(base + L.bounds);
            for (int64_t e = 0; e < neta; ++e)
                table[(size_t)e] = make_rev_job((const cplx*)vec_out + e * vec_stride, w_out + e, th_red + e * M, keep_n[e], t.g,
                                                etas[e], bounds + (size_t)e * kRevWords);
            // which curvatures sit on a uniform theta grid (every one, on the path: centres of linspace edges, cropped to a contiguous
            // run): their back-map is the diagonal kernel (thth.hip).  One kernel over the table, one read-back.
            SCINT_HIP(hipMemcpyAsync(base + L.jobs, table.data(), sizeof(RevJobDev) * (size_t)neta, hipMemcpyHostToDevice, st));
            rc = launch_rev_uniform((const RevJobDev*)(base + L.jobs), neta, t.g, (int32_t*)(base + L.uflags), st);
            if (rc != SCINT_OK) return rc;
            std::vector<int32_t> uflags((size_t)neta);
            SCINT_HIP(hipMemcpyAsync(uflags.data(), base + L.uflags, sizeof(int32_t) * (size_t)neta, hipMemcpyDeviceToHost, st));
            SCINT_HIP(hipStreamSynchronize(st));
            t.uniform.resize((size_t)neta);
            for (int64_t e = 0; e < neta; ++e) t.uniform[(size_t)e] = uflags[(size_t)e] != 0;
            // partner tables of the general back-map for the largest groups of curvatures that keep the same theta centres and are
            // NOT on a uniform grid (thth.hpp)
            if (crop_group) {
                std::map<int32_t, std::vector<int64_t>> groups;
                for (int64_t e = 0; e < neta; ++e)
                    if (crop_group[e] >= 0 && keep_n[e] >= 3 && !t.uniform[(size_t)e]) groups[crop_group[e]].push_back(e);
                std::vector<const std::vector<int64_t>*> big;
                for (const auto& kv : groups)
                    if ((int)kv.second.size() >= kRevWalkMinGroup) big.push_back(&kv.second);
                std::sort(big.begin(), big.end(), [](const std::vector<int64_t>* a, const std::vector<int64_t>* b) { return a->size() > b->size(); });
                // (1) the promise, checked: equal counts on the host, equal rows on the device (flags in the job table's space, which
                //     is filled after them; one small read-back)
                int32_t* flags_dev = (int32_t*)(base + L.jobs);
                int32_t flags[kRevWalkTables];
                size_t ntab = 0;
                for (size_t ti = 0; ti < big.size() && ti < (size_t)kRevWalkTables; ++ti) { flags[ti] = 1; ++ntab; }
                if (ntab) SCINT_HIP(hipMemcpyAsync(flags_dev, flags, sizeof(int32_t) * ntab, hipMemcpyHostToDevice, st));
                for (size_t ti = 0; ti < ntab; ++ti) {
                    const std::vector<int64_t>& members = *big[ti];
                    const int64_t n0 = keep_n[members[0]];
                    for (size_t m0 = 1; m0 < members.size(); m0 += 255) {
                        RowList rl;
                        rl.rows[0] = (int32_t)members[0];
                        int cnt = 1;
                        for (size_t m = m0; m < members.size() && cnt < 256; ++m) rl.rows[cnt++] = (int32_t)members[m];
                        hipLaunchKernelGGL(rows_equal_kernel, dim3((unsigned)cnt), dim3(256), 0, st, th_red, M, rl, (int)n0, flags_dev + ti);
                    }
                }
                if (ntab) {
                    SCINT_LAUNCH_CHECK();
                    SCINT_HIP(hipMemcpyAsync(flags, flags_dev, sizeof(int32_t) * ntab, hipMemcpyDeviceToHost, st));
                    SCINT_HIP(hipStreamSynchronize(st));
                }
                // (2) one table per verified group
                for (size_t ti = 0; ti < ntab; ++ti) {
                    const std::vector<int64_t>& members = *big[ti];
                    const int64_t e0 = members[0], n0 = keep_n[e0];
                    bool same = flags[ti] != 0;
                    for (int64_t e : members) same = same && keep_n[e] == n0;
                    if (!same) continue;                       // (its members walk in the kernel, as without a group)
                    uint8_t* masks = (uint8_t*)(base + L.walk[ti]);
                    uint8_t* col_ok = masks + (size_t)geom->nfd * (size_t)n0;
                    rc = launch_rev_walk_table(th_red + e0 * M, n0, t.g, masks, col_ok, st);
                    if (rc != SCINT_OK) return rc;
                    for (int64_t e : members) { table[(size_t)e].walk = masks; table[(size_t)e].walk_col = col_ok; }
                }
            }
            SCINT_HIP(hipMemcpyAsync(base + L.jobs, table.data(), sizeof(RevJobDev) * (size_t)neta, hipMemcpyHostToDevice, st));
            SCINT_HIP(hipStreamSynchronize(st));       // (the table is a local: the copy has left it)
            t.jobs_dev = (const RevJobDev*)(base + L.jobs);
            t.images = L.images;
        }
    }
    rc = run_sweep(cs, 1, 0, nullptr, geom, th_cents, M, keep_idx, keep_n, etas, neta, tol, max_iter, batch, w_out,
                   status_out, iters_out, true, (cplx*)vec_out, vec_stride, &t, base + L.sweep, L.sweep_bytes, stream);
    g_chisq_fused = t.fuse ? 1 : 0; g_chisq_redone = 0;
    if (rc != SCINT_OK || !t.fuse) return rc;
    // the curvatures whose back-map met a pair ON a bin edge (their histogram may not be mirror-symmetric): chi^2 once more, from the
    // written image (their eigenvectors are in vec_out); the sweep has drained its streams
    std::vector<int32_t> asym((size_t)neta);
    SCINT_HIP(hipMemcpyAsync(asym.data(), t.asym, sizeof(int32_t) * (size_t)neta, hipMemcpyDeviceToHost, st));
    SCINT_HIP(hipStreamSynchronize(st));
    t.fuse = false;
    std::vector<int64_t> again;
    for (int64_t e = 0; e < neta; ++e)
        if (asym[(size_t)e]) again.push_back(e);
    for (size_t k = 0; k < again.size(); k += (size_t)t.batch_max()) {
        const int cnt = (int)std::min<size_t>((size_t)t.batch_max(), again.size() - k);
        rc = t.retire_batch(again.data() + k, cnt, st, 0);
        if (rc != SCINT_OK) return rc;
    }
    if (!again.empty()) SCINT_HIP(hipStreamSynchronize(st));
    g_chisq_redone = (int64_t)again.size();
    return SCINT_OK;
}
