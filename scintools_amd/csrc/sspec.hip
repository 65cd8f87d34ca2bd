// sspec.hip -- Dynspec.calc_sspec (dynspec.py:3665-3721) in two HBM round trips.
//
// The secondary spectrum is |FFT2|^2 of the windowed, mean-subtracted REAL dynamic spectrum
// d'[nf, nt], zero-padded to R x C = 2 next_pow2(nf) x 2 next_pow2(nt) (always >= 2x: the upper half
// of either axis is zeros), of which only the delay rows k1 < R/2 are kept (`halve`).  The generic
// 2-D driver of fft.hip spends four trips through HBM on it (row transforms, two tiled passes along the
// strided axis of the half-width spectrum, 5.4x the algorithmic bytes; profiles/r02_pmc_sspec4096.json).
// Here the STRIDED axis goes first, on the real input, and both axes exploit the zero half:
//
//   an n2 = 2n point transform of a sequence whose upper half is zero splits into two INDEPENDENT
//   n-point transforms of the same n inputs:  X[2m] = FFT_n(x[s])[m],  X[2m+1] = FFT_n(x[s] W_2n^s)[m]
//   -- so a transform never needs more than n complex values in LDS, and a workgroup can do the two
//   halves one after the other (rows) or two workgroups can share the input (columns).
//
//   sspec_cols_kernel   axis 0 (frequency -> delay).  Two adjacent REAL columns c, c+1 ride as one complex
//                       sequence z[r] = d'[r, c] + i d'[r, c+1]; a slot (n/16 threads) transforms one such
//                       pair, a 1024-thread workgroup holds 16384 points: 4 adjacent pairs at n = 4096 (64
//                       contiguous bytes per input row, 128 contiguous bytes per output row).  The real
//                       spectra are separated from Z[k], Z[2n - k] (both inside the same half) and stored
//                       as Y[k1][c], k1 = 2m + half < R/2: exactly the non-redundant half, [R/2, nt] complex.
//                       Window, both means and the prewhitening stencil are fused into the loads.
//   sspec_rows_kernel   axis 1 (time -> Doppler) of the kept rows only: one row per slot, both halves in
//                       sequence, |.|^2, post-darkening, 10 log10 and the fftshift fused into 16-byte
//                       stores (the even / odd Doppler bins of one thread are neighbours in memory).
//
// HBM traffic at 4096^2: 134 MB (means) + 134..268 (input, twice through the L2 / Infinity Cache) + 268
// (Y out) + 268 (Y in) + 268 (dB out) = 1.07..1.2 GB for 0.40 GB algorithmic, against 2.18 GB before.
// Shapes outside 256 <= R/2, C/2 <= 8192 and `halve = 0` keep the generic path (fft.hip).
#include "sspec.hpp"

#include "fft.hpp"

namespace scint {

// d'(r, c) of dynspec.py:3667-3674 with NumPy's operation order (as WindowedValue / RowSource in fft.hip)
struct SspecIn {
    const double* dyn; const double* wt; const double* wf; const double* scal;   // scal[0] = mean 1, scal[1] = mean 2
    int nf, nt, nf_eff, nt_eff, prewhite;
};
__device__ inline double sspec_d(double x, double wt, double wf, double m1, double m2, bool windowed) {
    double v = x - m1;
    if (windowed) { v = wt * v; v = wf * v; }
    return v - m2;
}

struct SspecCols {
    SspecIn in;
    cplx* Y; int ldY;             // [R/2][ldY] complex, ldY = 2 * npairs
    int npairs;                   // ceil(nt_eff / 2) column pairs
    const cplx* tw_n;             // W_n
    const cplx* tw_2n;            // W_2n (the odd half's input twiddle)
    int xcd_remap;
};

// one Stockham exchange through n doubles of LDS per slot: real parts, then imaginary parts
template <int RP, int RN>
__device__ inline void split_exchange(cplx (&v)[kEPT], double* ldsd, int t, int Tr, int n, int Ns) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
        for (int m = 0; m < RP; ++m) ldsd[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m].x;
    __syncthreads();
    double re[kEPT];
#pragma unroll
    for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
        for (int m = 0; m < RN; ++m) re[q * RN + m] = ldsd[lds_pad(t + q * Tr + m * (n / RN))];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
        for (int m = 0; m < RP; ++m) ldsd[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m].y;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
        for (int m = 0; m < RN; ++m) v[q * RN + m] = mk(re[q * RN + m], ldsd[lds_pad(t + q * Tr + m * (n / RN))]);
}
template <int RP, int RN>
__device__ inline void full_exchange(cplx (&v)[kEPT], cplx* lds, int t, int Tr, int n, int Ns) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
        for (int m = 0; m < RP; ++m) lds[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
        for (int m = 0; m < RN; ++m) v[q * RN + m] = lds[lds_pad(t + q * Tr + m * (n / RN))];
}

// the n-point transform of the 16 values a thread holds (stage-0 order in, last-stage order out)
template <int R0, int R1, int R2, int R3, bool SPLIT>
__device__ inline void slot_fft(cplx (&v)[kEPT], void* lds, int t, const cplx* __restrict__ tw) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    stockham_compute<R0>(v, t, Tr, n, 1, tw);
    if constexpr (R1 > 1) {
        if constexpr (SPLIT) split_exchange<R0, R1>(v, (double*)lds, t, Tr, n, 1);
        else full_exchange<R0, R1>(v, (cplx*)lds, t, Tr, n, 1);
        stockham_compute<R1>(v, t, Tr, n, R0, tw);
        if constexpr (R2 > 1) {
            if constexpr (SPLIT) split_exchange<R1, R2>(v, (double*)lds, t, Tr, n, R0);
            else full_exchange<R1, R2>(v, (cplx*)lds, t, Tr, n, R0);
            stockham_compute<R2>(v, t, Tr, n, R0 * R1, tw);
            if constexpr (R3 > 1) {
                if constexpr (SPLIT) split_exchange<R2, R3>(v, (double*)lds, t, Tr, n, R0 * R1);
                else full_exchange<R2, R3>(v, (cplx*)lds, t, Tr, n, R0 * R1);
                stockham_compute<R3>(v, t, Tr, n, R0 * R1 * R2, tw);
            }
        }
    }
}
template <int R0, int R1, int R2, int R3>
struct LastStage {
    static constexpr int RL = (R3 > 1) ? R3 : (R2 > 1) ? R2 : (R1 > 1) ? R1 : R0;
    static constexpr int Ns = (R3 > 1) ? R0 * R1 * R2 : (R2 > 1) ? R0 * R1 : (R1 > 1) ? R0 : 1;
};

constexpr int kColsBlock = 1024;

template <int R0, int R1, int R2, int R3>
__global__ void __launch_bounds__(kColsBlock) sspec_cols_kernel(SspecCols a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT, SPB = kColsBlock / Tr;
    using LS = LastStage<R0, R1, R2, R3>;
    const int i = (int)threadIdx.x / Tr, t = (int)threadIdx.x - i * Tr;
    double* ldsd = reinterpret_cast<double*>(smem_raw) + (size_t)i * lds_pad(n);
    // consecutive workgroup ids land on different XCDs (own L2 each): give every XCD a contiguous range of
    // logical blocks, so that the two halves of a tile and its neighbours share input lines in one L2
    int lb = (int)blockIdx.x;
    if (a.xcd_remap) lb = (lb & 7) * ((int)gridDim.x >> 3) + (lb >> 3);
    const int tile = lb >> 1, half = lb & 1;
    const int p = tile * SPB + i;
    const bool active = p < a.npairs;
    const int c0 = 2 * p;
    const SspecIn& in = a.in;
    const bool windowed = in.wt != nullptr;
    const double m1 = in.scal[0], m2 = in.scal[1];
    const bool has1 = active && c0 + 1 < in.nt_eff;          // the pair's second column exists
    double w0 = 1.0, w1 = 1.0, w2 = 1.0;
    if (windowed && active) {
        w0 = in.wt[c0];
        w1 = c0 + 1 < in.nt ? in.wt[c0 + 1] : 0.0;
        w2 = c0 + 2 < in.nt ? in.wt[c0 + 2] : 0.0;
    }
    const bool aligned = (in.nt & 1) == 0;
    cplx v[kEPT];
#pragma unroll
    for (int q = 0; q < kEPT / R0; ++q) {
#pragma unroll
        for (int m = 0; m < R0; ++m) {
            const int s = t + q * Tr + m * (n / R0);          // input row
            cplx z = mk(0.0, 0.0);
            if (active && s < in.nf_eff) {
                const double* row = in.dyn + (int64_t)s * in.nt + c0;
                const double f0 = windowed ? in.wf[s] : 1.0;
                double x0, x1 = 0.0;
                if (aligned) { const v2d xx = *(const SCINT_GLOBAL v2d*)row; x0 = xx.x; x1 = xx.y; }
                else { x0 = row[0]; if (c0 + 1 < in.nt) x1 = row[1]; }
                const double d00 = sspec_d(x0, w0, f0, m1, m2, windowed);
                const double d01 = sspec_d(x1, w1, f0, m1, m2, windowed);
                if (!in.prewhite) {
                    z = mk(d00, has1 ? d01 : 0.0);
                } else {
                    // convolve2d([[1,-1],[-1,1]], d', 'valid')  (dynspec.py:3681): pw[r, c] =
                    // d'[r+1, c+1] - d'[r+1, c] - d'[r, c+1] + d'[r, c], same association as fft.hip
                    const double f1 = windowed ? in.wf[s + 1] : 1.0;
                    const double* rown = row + in.nt;
                    double y0, y1 = 0.0;
                    if (aligned) { const v2d yy = *(const SCINT_GLOBAL v2d*)rown; y0 = yy.x; y1 = yy.y; }
                    else { y0 = rown[0]; if (c0 + 1 < in.nt) y1 = rown[1]; }
                    const double d10 = sspec_d(y0, w0, f1, m1, m2, windowed);
                    const double d11 = sspec_d(y1, w1, f1, m1, m2, windowed);
                    double zy = 0.0;
                    if (has1) {
                        const double d02 = sspec_d(row[2], w2, f0, m1, m2, windowed);
                        const double d12 = sspec_d(rown[2], w2, f1, m1, m2, windowed);
                        zy = d12 - d11 - d02 + d01;
                    }
                    z = mk(d11 - d10 - d01 + d00, zy);
                }
                if (half) z = z * a.tw_2n[s];
            }
            v[q * R0 + m] = z;
        }
    }
    slot_fft<R0, R1, R2, R3, true>(v, ldsd, t, a.tw_n);
    // natural order in LDS (real parts, then imaginary parts); thread t separates the two real spectra
    // at m = t + k Tr < n/2 from Z[m] and its partner Z[n - m] (even half) / Z[n - 1 - m] (odd half)
    double are[8], bre[8];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / LS::RL; ++q)
#pragma unroll
        for (int m = 0; m < LS::RL; ++m)
            ldsd[lds_pad(stockham_out_index<LS::RL>(t, Tr, LS::Ns, q, m))] = v[q * LS::RL + m].x;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int m = t + k * Tr, pm = half ? n - 1 - m : (n - m) & (n - 1);
        are[k] = ldsd[lds_pad(m)];
        bre[k] = ldsd[lds_pad(pm)];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kEPT / LS::RL; ++q)
#pragma unroll
        for (int m = 0; m < LS::RL; ++m)
            ldsd[lds_pad(stockham_out_index<LS::RL>(t, Tr, LS::Ns, q, m))] = v[q * LS::RL + m].y;
    __syncthreads();
    if (!active) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int m = t + k * Tr, pm = half ? n - 1 - m : (n - m) & (n - 1);
        const double aim = ldsd[lds_pad(m)], bim = ldsd[lds_pad(pm)];
        // X_c = (Z[k] + conj Z[-k]) / 2,  X_{c+1} = (Z[k] - conj Z[-k]) / (2i)
        cplx* out = a.Y + (int64_t)(2 * m + half) * a.ldY + c0;
        gstore(out, mk(0.5 * (are[k] + bre[k]), 0.5 * (aim - bim)));
        gstore(out + 1, mk(0.5 * (aim + bim), -0.5 * (are[k] - bre[k])));
    }
}

struct SspecRows {
    const cplx* Y; int ldY; int nt_eff;
    int nrows;                    // R/2 kept delay rows
    int C;                        // 2n
    const cplx* tw_n; const cplx* tw_2n;
    double* out;                  // [R/2][C] dB
    int prewhite; const double* pd_fd; const double* pd_td;
};

// KEEP: the 16 inputs of a thread stay in registers for the second half (else they are loaded again).
template <int R0, int R1, int R2, int R3, bool SPLIT, bool KEEP>
__global__ void __launch_bounds__((R0 * R1 * R2 * R3 / kEPT) >= 256 ? (R0 * R1 * R2 * R3 / kEPT) : 256)
sspec_rows_kernel(SspecRows a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    using LS = LastStage<R0, R1, R2, R3>;
    const int i = (int)threadIdx.x / Tr, t = (int)threadIdx.x - i * Tr;
    const int spb = (int)blockDim.x / Tr;
    const int k1 = (int)blockIdx.x * spb + i;
    const bool active = k1 < a.nrows;
    void* lds = SPLIT ? (void*)(reinterpret_cast<double*>(smem_raw) + (size_t)i * lds_pad(n))
                      : (void*)(reinterpret_cast<cplx*>(smem_raw) + (size_t)i * lds_pad(n));
    const cplx* __restrict__ row = a.Y + (int64_t)(active ? k1 : 0) * a.ldY;
    cplx x[KEEP ? kEPT : 1];
    double pe[kEPT];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        cplx v[kEPT];
#pragma unroll
        for (int q = 0; q < kEPT / R0; ++q) {
#pragma unroll
            for (int m = 0; m < R0; ++m) {
                const int s = t + q * Tr + m * (n / R0);
                cplx z;
                if (KEEP && half == 1) z = x[KEEP ? q * R0 + m : 0];
                else z = (active && s < a.nt_eff) ? gload(row + s) : mk(0.0, 0.0);
                if (KEEP && half == 0) x[KEEP ? q * R0 + m : 0] = z;
                if (half == 1) z = z * a.tw_2n[s];
                v[q * R0 + m] = z;
            }
        }
        slot_fft<R0, R1, R2, R3, SPLIT>(v, lds, t, a.tw_n);
        if (half == 0) {
#pragma unroll
            for (int e = 0; e < kEPT; ++e) pe[e] = v[e].x * v[e].x + v[e].y * v[e].y;
            __syncthreads();      // the first half's last exchange is read before the second half's first one writes
        } else if (active) {
            const double td = a.prewhite ? a.pd_td[k1] : 1.0;
            double* __restrict__ orow = a.out + (int64_t)k1 * a.C;
#pragma unroll
            for (int q = 0; q < kEPT / LS::RL; ++q) {
#pragma unroll
                for (int m = 0; m < LS::RL; ++m) {
                    const int e = q * LS::RL + m;
                    const int mm = stockham_out_index<LS::RL>(t, Tr, LS::Ns, q, m);
                    // Doppler bins 2 mm and 2 mm + 1 at their fftshift-ed place (dynspec.py:3687); C/2 is even
                    const int col = (2 * mm + n) & (a.C - 1);
                    double p0 = pe[e], p1 = v[e].x * v[e].x + v[e].y * v[e].y;
                    if (a.prewhite) {   // post-darkening, column C/2 and row 0 forced to 1 (dynspec.py:3704-3717)
                        const double d0 = (col == n || k1 == 0) ? 1.0 : a.pd_fd[col] * td;
                        const double d1 = (k1 == 0) ? 1.0 : a.pd_fd[col + 1] * td;
                        p0 = p0 / d0; p1 = p1 / d1;
                    }
                    v2d o; o.x = 10.0 * log10(p0); o.y = 10.0 * log10(p1);
                    __builtin_nontemporal_store(o, (SCINT_GLOBAL v2d*)(orow + col));
                }
            }
        }
    }
}

template <int R0, int R1, int R2, int R3>
static int32_t launch_cols(const SspecCols& a, hipStream_t stream) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT, SPB = kColsBlock / Tr;
    const int tiles = (int)ceil_div(a.npairs, SPB), grid = 2 * tiles;
    SspecCols b = a;
    b.xcd_remap = (grid % 8 == 0) ? 1 : 0;
    const size_t lds = (size_t)SPB * (size_t)(n + n / 16) * sizeof(double);
    auto k = sspec_cols_kernel<R0, R1, R2, R3>;
    SCINT_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kColsBlock), lds, stream, b);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

template <int R0, int R1, int R2, int R3>
static int32_t launch_rows(const SspecRows& a, hipStream_t stream) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    const int block = Tr >= 256 ? Tr : 256, spb = block / Tr;
    const int grid = (int)ceil_div(a.nrows, spb);
    // n <= 4096: the whole transform in LDS as complex values (68 KiB at 4096: two workgroups per CU, 256
    // registers each -- the inputs stay in registers for the second half); 8192: real / imaginary parts
    // in turn (68 KiB), inputs loaded again
    if constexpr (n <= 4096) {
        const size_t lds = (size_t)spb * (size_t)(n + n / 16) * sizeof(cplx);
        auto k = sspec_rows_kernel<R0, R1, R2, R3, false, true>;
        if (lds > 64 * 1024)
            SCINT_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, a);
    } else {
        const size_t lds = (size_t)spb * (size_t)(n + n / 16) * sizeof(double);
        auto k = sspec_rows_kernel<R0, R1, R2, R3, true, false>;
        if (lds > 64 * 1024)
            SCINT_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, a);
    }
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

#define SCINT_SSPEC_DISPATCH(N, F, ...)                                  \
    switch (N) {                                                         \
        case 256: return F<16, 16, 1, 1>(__VA_ARGS__);                   \
        case 512: return F<16, 16, 2, 1>(__VA_ARGS__);                   \
        case 1024: return F<16, 16, 4, 1>(__VA_ARGS__);                  \
        case 2048: return F<16, 16, 8, 1>(__VA_ARGS__);                  \
        case 4096: return F<16, 16, 16, 1>(__VA_ARGS__);                 \
        case 8192: return F<16, 16, 16, 2>(__VA_ARGS__);                 \
        default: break;                                                  \
    }
static int32_t dispatch_cols(int64_t n, const SspecCols& a, hipStream_t s) {
    SCINT_SSPEC_DISPATCH(n, launch_cols, a, s)
    SCINT_REQUIRE(false, "sspec: unsupported column transform length");
}
static int32_t dispatch_rows(int64_t n, const SspecRows& a, hipStream_t s) {
    SCINT_SSPEC_DISPATCH(n, launch_rows, a, s)
    SCINT_REQUIRE(false, "sspec: unsupported row transform length");
}

bool sspec_fast_supported(int64_t nf, int64_t nt, int32_t halve) {
    if (!halve || nf < 3 || nt < 3) return false;
    static const int off = [] { const char* e = getenv("SCINT_SSPEC_GENERIC"); return e ? atoi(e) : 0; }();
    if (off) return false;        // tests compare the two paths
    const int64_t nr = next_pow2(nf), nc = next_pow2(nt);   // R/2, C/2
    return nr >= 256 && nr <= 8192 && nc >= 256 && nc <= 8192;
}

size_t sspec_fast_workspace(int64_t nf, int64_t nt) {
    const int64_t nr = next_pow2(nf);
    return align_up(sizeof(cplx) * (size_t)nr * (size_t)(2 * ceil_div(nt, 2)), 256);
}

int32_t sspec_fast(const double* dyn, int64_t nf, int64_t nt, const double* win_t, const double* win_f,
                   const double* scal, int32_t prewhite, const double* pd_fd, const double* pd_td,
                   double* sec_out, void* workspace, hipStream_t stream) {
    const int64_t nr = next_pow2(nf), nc = next_pow2(nt);
    const int64_t nf_eff = prewhite ? nf - 1 : nf, nt_eff = prewhite ? nt - 1 : nt;
    const cplx* tw_r = twiddle_table(nr);
    const cplx* tw_2r = twiddle_table(2 * nr);
    const cplx* tw_c = twiddle_table(nc);
    const cplx* tw_2c = twiddle_table(2 * nc);
    if (!tw_r || !tw_2r || !tw_c || !tw_2c) return SCINT_E_HIP;
    SspecCols ca{};
    ca.in = SspecIn{dyn, win_t, win_f, scal, (int)nf, (int)nt, (int)nf_eff, (int)nt_eff, prewhite};
    ca.npairs = (int)ceil_div(nt_eff, 2);
    ca.Y = (cplx*)workspace; ca.ldY = 2 * ca.npairs;
    ca.tw_n = tw_r; ca.tw_2n = tw_2r;
    int32_t rc = dispatch_cols(nr, ca, stream);
    if (rc != SCINT_OK) return rc;
    SspecRows ra{};
    ra.Y = ca.Y; ra.ldY = ca.ldY; ra.nt_eff = (int)nt_eff; ra.nrows = (int)nr; ra.C = (int)(2 * nc);
    ra.tw_n = tw_c; ra.tw_2n = tw_2c; ra.out = sec_out;
    ra.prewhite = prewhite; ra.pd_fd = pd_fd; ra.pd_td = pd_td;
    return dispatch_rows(nc, ra, stream);
}

}  // namespace scint
