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
//   sspec_prep_kernel   pass 0: the sums behind both means AND a pair-major copy of the input, transposed
//                       through LDS (the column kernel then streams its two real columns contiguously).
//   sspec_cols2_kernel  axis 0 (frequency -> delay), persistent.  Two adjacent REAL columns c, c+1 ride as one complex
//                       sequence z[r] = d'[r, c] + i d'[r, c+1]; a workgroup (n/16 threads) walks over such pairs and
//                       does BOTH halves of a pair from one read of it, the next pair on its way through the second
//                       transform.  The real spectra are separated from Z[k], Z[2n - k] (both inside the same half)
//                       and stored as Y[k1][c], k1 = 2m + half < R/2: exactly the non-redundant half, [R/2, nt]
//                       complex, in tiles of (4 rows of one parity) x (pair) = 128 bytes, so that four consecutive
//                       threads store one whole line.  Window and both means are fused into the loads.
//                       (sspec_cols_kernel: round 5's one-shot form, one (pair, half) per workgroup; it keeps the
//                       prewhitening stencil, whose three extra loads per value are not prefetched.)
//   sspec_rows2_kernel  axis 1 (time -> Doppler) of the kept rows only, persistent: both halves of a row from one read,
//                       |.|^2, post-darkening, 10 log10 (table form, branch-free in groups) and the fftshift fused into
//                       16-byte stores of (even bin, odd bin) pairs -- whole lines.
//
// HBM traffic at 4096^2: 134 MB in + 134 (pair-major copy out) + 134 (in) + 268 (Y out) + 268 (Y in) + 268 (dB out) = 1.2 GB
// for 0.40 GB algorithmic (2.18 GB with the generic driver).  Round 6 (profiles/r06_*): the persistent kernels read every
// row / pair once (round 5's read them once per half), store whole lines, keep their twiddles in registers and hide the load
// latency behind the second transform: 0.443 -> 0.38 ms at 4096^2, 1.85 -> 1.55 ms at 8192^2.  What the kernels' own clocks
// say is left (profiles/r06_rows2_phase_times*.txt, r06_cols2_phase_times.txt, r06_rows2_ablation.txt): the column kernel's
// steady state moves 192 KB per pair and workgroup in 41 600 clocks -- 9 B per clock and CU, the chip's HBM rate -- and pays
// its four-iteration ramp; the row kernel's waves stand at the memory-issue port a third of a row's 38 000 clocks (sixteen
// loads of 32 bytes out of every 128-byte line of the 4-row tiles: 2048 L1 misses a row; row-major Y instead costs the column
// kernel twice that in partial-line stores, profiles/r06_sspec_rowmajor_wgs_ab.txt), its arithmetic alone is 80 us, arithmetic +
// exchanges 115, everything 205 -- two workgroups per CU (registers: 256 each) overlap little of it.
// Shapes outside 256 <= R/2, C/2 <= 8192 and `halve = 0` keep the generic path (fft.hip).
#include "sspec.hpp"
#include "prof.hpp"

#include "fft.hpp"

#ifndef SCINT_ROWS2_LOG_GROUP
#define SCINT_ROWS2_LOG_GROUP 4
#endif
#ifndef SCINT_ROWS2_LOAD_PIECES
#define SCINT_ROWS2_LOAD_PIECES 4
#endif
#ifndef SCINT_ROWS2_STORE_PIECES
#define SCINT_ROWS2_STORE_PIECES 1
#endif

namespace scint {

// d'(r, c) of dynspec.py:3667-3674 with NumPy's operation order (as WindowedValue / RowSource in fft.hip)
struct SspecIn {
    const double* dynp;           // pair-major copy of the dynamic spectrum: [ceil(nt/2)][nf][2] (sspec_prep_kernel)
    const double* wt; const double* wf; const double* scal;   // scal[0] = mean 1, scal[1] = mean 2
    int nf, nt, nf_eff, nt_eff, prewhite;
    const double* partial; int npartial;                      // sspec_prep_kernel's per-tile sums (the persistent column kernel adds them itself)
};

// Pass 0: one read of the dynamic spectrum gives (a) the three sums behind both means of dynspec.py:3667-3674
// (m1 = mean(dyn), m2 = mean(w_f w_t (dyn - m1)) = (S_wd - m1 S_w) / n, fixed-order partials per 64 x 64 tile) and
// (b) a PAIR-MAJOR copy dynp[c / 2][r][c % 2]: the column kernel then reads its two real columns as one
// contiguous stream of 16-byte elements instead of 16 bytes out of every row (one 128-byte line each; that
// cost it 106 of 283 us at 4096^2, profiles/r03_sspec_ablation.txt).  Tiles transpose through LDS: reads are
// 512-byte row segments, writes 1-KiB runs of one pair.
__global__ void __launch_bounds__(256) sspec_prep_kernel(const double* __restrict__ dyn, const double* __restrict__ wt,
                                                         const double* __restrict__ wf, int nf, int nt,
                                                         double* __restrict__ dynp, double* __restrict__ partial) {
    __shared__ double tile[64][65];
    __shared__ double red[4];
    const int c0 = (int)blockIdx.x * 64, r0 = (int)blockIdx.y * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double sd = 0.0, swd = 0.0, sw = 0.0;
    if ((nt & 1) == 0 && ((uintptr_t)dyn & 15) == 0) {
        // even row length (and an aligned plane): every pair of adjacent columns is one aligned 16-byte load (a wave instruction reads two 512-byte row
        // segments; half the load instructions of the scalar form below -- round 4: the copy ran at 4.1 TB/s where a plain
        // copy of the same bytes reaches 7 with the Infinity Cache's help)
        const int cp = lane & 31, c = c0 + 2 * cp, rsub = (lane >> 5) + 2 * w;
        const bool in = c < nt;                                   // (nt even: c + 1 < nt too)
        const double wc0 = (wt && in) ? wt[c] : 1.0, wc1 = (wt && in) ? wt[c + 1] : 1.0;
#pragma unroll 4
        for (int i = 0; i < 8; ++i) {
            const int rr = rsub + 8 * i, r = r0 + rr;
            v2d d; d.x = 0.0; d.y = 0.0;
            if (r < nf && in) {
                d = *(const SCINT_GLOBAL v2d*)(dyn + (int64_t)r * nt + c);
                const double f = wf ? wf[r] : 1.0;
                const double w0 = wf ? f * wc0 : 1.0, w1 = wf ? f * wc1 : 1.0;
                sd += d.x; swd += w0 * d.x; sw += w0;
                sd += d.y; swd += w1 * d.y; sw += w1;
            }
            tile[rr][2 * cp] = d.x; tile[rr][2 * cp + 1] = d.y;
        }
    } else {
        const int c = c0 + lane;
        const double wc = (wt && c < nt) ? wt[c] : 1.0;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int rr = w + 4 * i, r = r0 + rr;
            double d = 0.0;
            if (r < nf && c < nt) {
                d = dyn[(int64_t)r * nt + c];
                const double ww = wf ? wf[r] * wc : 1.0;
                sd += d; swd += ww * d; sw += ww;
            }
            tile[rr][lane] = d;
        }
    }
    __syncthreads();
    // pair pp of the tile, row `lane`: 16 bytes per lane, 1 KiB per wave instruction
    const int npair = (nt + 1) / 2;
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
        const int pp = w + 4 * i, p = (c0 >> 1) + pp, r = r0 + lane;
        if (p < npair && r < nf) {
            v2d o; o.x = tile[lane][2 * pp]; o.y = tile[lane][2 * pp + 1];
            *(SCINT_GLOBAL v2d*)(dynp + ((int64_t)p * nf + r) * 2) = o;
        }
    }
    const int b = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, nb = (int)(gridDim.x * gridDim.y);
    sd = block_sum(sd, red); swd = block_sum(swd, red); sw = block_sum(sw, red);
    if (threadIdx.x == 0) { partial[b] = sd; partial[nb + b] = swd; partial[2 * nb + b] = sw; }
}
__global__ void __launch_bounds__(256) sspec_prep_means_kernel(const double* partial, int np, double n, double* scal) {
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

// Workgroup barrier of the transforms.  LB = true: wait for the LDS counter only (lds_barrier) -- a persistent
// kernel keeps the loads of its NEXT row / column pair in flight across the barriers of this one's transform;
// __syncthreads() would drain them (its fence waits for vmcnt(0)).
template <bool LB> __device__ inline void xbarrier() { if constexpr (LB) lds_barrier(); else __syncthreads(); }

// one Stockham exchange through n doubles of LDS per slot: real parts, then imaginary parts
template <int RP, int RN, bool LB = false>
__device__ inline void split_exchange(cplx (&v)[kEPT], double* ldsd, int t, int Tr, int n, int Ns) {
    xbarrier<LB>();
#pragma unroll
    for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
        for (int m = 0; m < RP; ++m) ldsd[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m].x;
    xbarrier<LB>();
    double re[kEPT];
#pragma unroll
    for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
        for (int m = 0; m < RN; ++m) re[q * RN + m] = ldsd[lds_pad(t + q * Tr + m * (n / RN))];
    xbarrier<LB>();
#pragma unroll
    for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
        for (int m = 0; m < RP; ++m) ldsd[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m].y;
    xbarrier<LB>();
#pragma unroll
    for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
        for (int m = 0; m < RN; ++m) v[q * RN + m] = mk(re[q * RN + m], ldsd[lds_pad(t + q * Tr + m * (n / RN))]);
}
// the n-point transform of the 16 values a thread holds (stage-0 order in, last-stage order out)
template <int R0, int R1, int R2, int R3, bool LB = false>
__device__ inline void slot_fft(cplx (&v)[kEPT], double* lds, int t, const cplx* __restrict__ tw) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    stockham_compute<R0>(v, t, Tr, n, 1, tw);
    if constexpr (R1 > 1) {
        split_exchange<R0, R1, LB>(v, lds, t, Tr, n, 1);
        stockham_compute<R1>(v, t, Tr, n, R0, tw);
        if constexpr (R2 > 1) {
            split_exchange<R1, R2, LB>(v, lds, t, Tr, n, R0);
            stockham_compute<R2>(v, t, Tr, n, R0 * R1, tw);
            if constexpr (R3 > 1) {
                split_exchange<R2, R3, LB>(v, lds, t, Tr, n, R0 * R1);
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

// The transforms of the persistent kernels take their stage twiddles from REGISTERS: a table read inside the transform would sit
// behind the prefetched loads of the next row on the in-order vmcnt counter and drain them (s_waitcnt vmcnt(0) at the first
// twiddle).  One table entry per thread and stage, read once per kernel: stage (Ns, R), butterfly q of the thread, needs
// W_{Ns R}^k, k = (t + q n/16) mod Ns -- in the stages before the last q = 0 only; in the last stage Ns R = n and
// W_n^(t + q n/16) = W_n^t W_16^q, the second factor a constant.
template <int R>
__device__ inline cplx stage_w(int t, int n, int Ns, const cplx* __restrict__ tw) {
    return tw[(t & (Ns - 1)) * (n / (Ns * R))];
}
template <int R>
__device__ inline void stockham_compute_w(cplx (&v)[kEPT], cplx w0in) {
    // (opaque: the powers below are loop-invariant in a persistent kernel, and the compiler would keep all fifteen of every
    //  stage in registers across the loop)
    double w0x = w0in.x, w0y = w0in.y;
    asm volatile("" : "+v"(w0x));
    asm volatile("" : "+v"(w0y));
    const cplx w0 = mk(w0x, w0y);
    static_for<0, kEPT / R>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        cplx x[R];
#pragma unroll
        for (int m = 0; m < R; ++m) x[m] = v[q * R + m];
        const cplx w1 = mul_w32<2 * q>(w0);
        if constexpr (R >= 4) {                        // the powers by multiplication, two interleaved chains (stockham_compute)
            const cplx w2 = mk(w1.x * w1.x - w1.y * w1.y, 2.0 * (w1.x * w1.y));
            cplx wo = w1, we = w2;
            x[1] = x[1] * wo;
            x[2] = x[2] * we;
#pragma unroll
            for (int m = 3; m < R; m += 2) {
                wo = wo * w2;
                x[m] = x[m] * wo;
                if (m + 1 < R) { we = we * w2; x[m + 1] = x[m + 1] * we; }
            }
        } else {
            x[1] = x[1] * w1;
        }
        SmallFFT<R>::run(x);
#pragma unroll
        for (int m = 0; m < R; ++m) v[q * R + m] = x[m];
    });
}
template <int R0, int R1, int R2, int R3>
struct StageW {
    static_assert(R0 == kEPT && R1 == kEPT && (R3 == 1 || R2 == kEPT), "only the last stage may hold several butterflies per thread");
    cplx w[3];
    __device__ inline void load(int t, const cplx* __restrict__ tw) {
        constexpr int n = R0 * R1 * R2 * R3;
        w[0] = stage_w<R1>(t, n, R0, tw);
        w[1] = R2 > 1 ? stage_w<R2>(t, n, R0 * R1, tw) : mk(1.0, 0.0);
        w[2] = R3 > 1 ? stage_w<R3>(t, n, R0 * R1 * R2, tw) : mk(1.0, 0.0);
    }
};
// slot_fft with the twiddles of StageW and LDS-only barriers.  `hook(c)` runs before the first stage (c = 0) and after every
// stage (c = 1 ..): the persistent kernels issue their memory instructions there in pieces -- a wave that issues sixteen loads
// (or stores) in one go stands at the issue port for as long as the memory pipeline takes to accept them (4000 - 11000 clocks
// a row, measured: profiles/r06_rows2_phase_times.txt), pieces drain while the next stage computes.
struct NoHook { template <class C> __device__ inline void operator()(C) const {} };
template <int R0, int R1, int R2, int R3> struct HookCount { static constexpr int value = 2 + (R2 > 1) + (R3 > 1) + 1; };
template <int R0, int R1, int R2, int R3, class Hook = NoHook>
__device__ inline void slot_fft_w(cplx (&v)[kEPT], double* lds, int t, const StageW<R0, R1, R2, R3>& sw, Hook&& hook = Hook()) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    hook(std::integral_constant<int, 0>{});
    stockham_compute<R0>(v, t, Tr, n, 1, nullptr);
    hook(std::integral_constant<int, 1>{});
    split_exchange<R0, R1, true>(v, lds, t, Tr, n, 1);
    stockham_compute_w<R1>(v, sw.w[0]);
    hook(std::integral_constant<int, 2>{});
    if constexpr (R2 > 1) {
        split_exchange<R1, R2, true>(v, lds, t, Tr, n, R0);
        stockham_compute_w<R2>(v, sw.w[1]);
        hook(std::integral_constant<int, 3>{});
        if constexpr (R3 > 1) {
            split_exchange<R2, R3, true>(v, lds, t, Tr, n, R0 * R1);
            stockham_compute_w<R3>(v, sw.w[2]);
            hook(std::integral_constant<int, 4>{});
        }
    }
}

// ... and with whole complex values through LDS (n complex per slot, two barriers per exchange instead of four): for a kernel whose
// workgroups per CU are set by registers, not by LDS
template <int RP, int RN>
__device__ inline void cplx_exchange(cplx (&v)[kEPT], cplx* lds, int t, int Tr, int n, int Ns) {
    lds_barrier();
#pragma unroll
    for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
        for (int m = 0; m < RP; ++m) lds[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m];
    lds_barrier();
#pragma unroll
    for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
        for (int m = 0; m < RN; ++m) v[q * RN + m] = lds[lds_pad(t + q * Tr + m * (n / RN))];
}
template <int R0, int R1, int R2, int R3>
__device__ inline void slot_fft_wc(cplx (&v)[kEPT], cplx* lds, int t, const StageW<R0, R1, R2, R3>& sw) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    stockham_compute<R0>(v, t, Tr, n, 1, nullptr);
    cplx_exchange<R0, R1>(v, lds, t, Tr, n, 1);
    stockham_compute_w<R1>(v, sw.w[0]);
    if constexpr (R2 > 1) {
        cplx_exchange<R1, R2>(v, lds, t, Tr, n, R0);
        stockham_compute_w<R2>(v, sw.w[1]);
        if constexpr (R3 > 1) {
            cplx_exchange<R2, R3>(v, lds, t, Tr, n, R0 * R1);
            stockham_compute_w<R3>(v, sw.w[2]);
        }
    }
}

// Workgroup of the column kernel: 256 threads (one pair at n = 4096, four workgroups and sixteen independent
// 4-wave barriers per CU).  Measured against 1024-thread workgroups holding four adjacent pairs (one
// 16-wave barrier domain per CU): 370 us -> see profiles/r03_fft_kernel_stats.csv.  The neighbours that
// complete a workgroup's 128-byte lines run beside it on the same XCD (block remap below) and meet in its L2.
template <int N> struct ColsBlock { static constexpr int value = (N / kEPT) >= 256 ? (N / kEPT) : 256; };

template <int R0, int R1, int R2, int R3>
__global__ void __launch_bounds__(ColsBlock<R0 * R1 * R2 * R3>::value, (ColsBlock<R0 * R1 * R2 * R3>::value >= 512 ? 2 : 4))   // (8192 points: 512 threads and 68 KiB of LDS -- two workgroups per CU whatever the bound says; 4 capped it at 64 registers and spilled)
sspec_cols_kernel(SspecCols a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT, kColsBlock = ColsBlock<n>::value, SPB = kColsBlock / Tr;
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
    const double* __restrict__ colp = in.dynp + (int64_t)p * in.nf * 2;        // this pair: [nf][2], contiguous
    const double* __restrict__ coln = colp + (int64_t)in.nf * 2;                // the next pair (prewhitening stencil)
    cplx v[kEPT];
#pragma unroll
    for (int q = 0; q < kEPT / R0; ++q) {
#pragma unroll
        for (int m = 0; m < R0; ++m) {
            const int s = t + q * Tr + m * (n / R0);          // input row
            cplx z = mk(0.0, 0.0);
            if (active && s < in.nf_eff) {
                const double f0 = windowed ? in.wf[s] : 1.0;
                const v2d xx = *(const SCINT_GLOBAL v2d*)(colp + 2 * s);
                const double d00 = sspec_d(xx.x, w0, f0, m1, m2, windowed);
                const double d01 = sspec_d(xx.y, w1, f0, m1, m2, windowed);
                if (!in.prewhite) {
                    z = mk(d00, has1 ? d01 : 0.0);
                } else {
                    // convolve2d([[1,-1],[-1,1]], d', 'valid')  (dynspec.py:3681): pw[r, c] =
                    // d'[r+1, c+1] - d'[r+1, c] - d'[r, c+1] + d'[r, c], same association as fft.hip
                    const double f1 = windowed ? in.wf[s + 1] : 1.0;
                    const v2d yy = *(const SCINT_GLOBAL v2d*)(colp + 2 * (s + 1));
                    const double d10 = sspec_d(yy.x, w0, f1, m1, m2, windowed);
                    const double d11 = sspec_d(yy.y, w1, f1, m1, m2, windowed);
                    double zy = 0.0;
                    if (has1) {
                        const double d02 = sspec_d(coln[2 * s], w2, f0, m1, m2, windowed);
                        const double d12 = sspec_d(coln[2 * (s + 1)], w2, f1, m1, m2, windowed);
                        zy = d12 - d11 - d02 + d01;
                    }
                    z = mk(d11 - d10 - d01 + d00, zy);
                }
                if (half) z = z * a.tw_2n[s];
            }
            v[q * R0 + m] = z;
        }
    }
    slot_fft<R0, R1, R2, R3>(v, ldsd, t, a.tw_n);
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
        // Y in tiles of (4 delay rows of one parity) x (one pair): 128 contiguous bytes from 4 consecutive
        // threads -- whole lines; row-major Y took 32 bytes per line per workgroup and 146 of the kernel's
        // 283 us (profiles/r03_sspec_ablation.txt)
        cplx* out = a.Y + ((int64_t)(m >> 2) * a.npairs + p) * 16 + half * 8 + (m & 3) * 2;
        gstore(out, mk(0.5 * (are[k] + bre[k]), 0.5 * (aim - bim)));
        gstore(out + 1, mk(0.5 * (aim + bim), -0.5 * (are[k] - bre[k])));
    }
}

// Round 6: the strided-axis pass as a persistent, self-pipelined kernel (the reasoning is the row kernel's, sspec_rows2_kernel below):
// a workgroup walks over column pairs and does BOTH halves of a pair from one read of it -- the windowed values stay in registers
// through the even half's transform, the odd half takes them times W_2n^s = W_2n^t W_32^m, and the registers then receive the next
// pair while the second transform runs.  (With the prewhitening stencil -- three more loads per value -- round 5's kernel above runs.)
template <int R0, int R1, int R2, int R3>
__global__ void __launch_bounds__(ColsBlock<R0 * R1 * R2 * R3>::value, 2)
sspec_cols2_kernel(SspecCols a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT, kColsBlock = ColsBlock<n>::value, SPB = kColsBlock / Tr;
    static_assert(R0 == kEPT, "one radix-16 butterfly per thread in stage 0: input s = t + m n/16");
    using LS = LastStage<R0, R1, R2, R3>;
    const int i = (int)threadIdx.x / Tr, t = (int)threadIdx.x - i * Tr, G = (int)gridDim.x;
    cplx* ldsc = reinterpret_cast<cplx*>(smem_raw) + (size_t)i * lds_pad(n);
    int lb = (int)blockIdx.x;
    if (a.xcd_remap) lb = (lb & 7) * (G >> 3) + (lb >> 3);
    const SspecIn& in = a.in;
    const bool windowed = in.wt != nullptr;
    // Both means from the per-tile sums of sspec_prep_kernel, by every workgroup for itself (the same sums in the same order: the
    // same bits everywhere) while its first pair is on the way -- a 7-us single-workgroup kernel and its launch otherwise
    double m1, m2;
    {
        __shared__ double red[8];
        double sd = 0.0, swd = 0.0, sw = 0.0;
        for (int j = threadIdx.x; j < in.npartial; j += kColsBlock) {
            sd += in.partial[j]; swd += in.partial[in.npartial + j]; sw += in.partial[2 * in.npartial + j];
        }
        sd = block_sum(sd, red); swd = block_sum(swd, red); sw = block_sum(sw, red);
        const double cnt = (double)in.nf * (double)in.nt;
        m1 = sd / cnt;
        m2 = (swd - m1 * sw) / cnt;
    }
    const cplx wt2 = a.tw_2n[t];                                  // W_2n^t
    StageW<R0, R1, R2, R3> sw;
    sw.load(t, a.tw_n);
    const int ntiles = (a.npairs + SPB - 1) / SPB;
    // the raw pair [nf][2] of pair p at this thread's rows (x = column 2 p, y = column 2 p + 1), and its two time-window values;
    // unconditional loads through a buffer resource: one per-lane byte offset, sixteen scalar distances (see the row kernel)
    const bool full = in.nf_eff == n;
    const int npairs_in = (in.nt + 1) / 2;                        // pairs of the pair-major copy
    const BufRsrc dres = make_rsrc(in.dynp, (int64_t)npairs_in * in.nf * (int64_t)sizeof(cplx));
    const BufRsrc fres = make_rsrc(in.wf, (int64_t)in.nf * (int64_t)sizeof(double));
    constexpr int kStep = Tr * (int)sizeof(cplx);
    // (no select between a load and its first use in the next iteration: values beyond a short column, and the time window beyond
    //  the last sample, are zeroed where they are consumed)
    auto load_pair = [&](int p, int tl, cplx (&r)[kEPT], double& w0, double& w1) {
        const int pp = p < a.npairs ? p : 0;
        if (windowed) { w0 = in.wt[2 * pp]; w1 = in.wt[2 * pp + 1 < in.nt ? 2 * pp + 1 : 2 * pp]; }
        const int voff = (pp * in.nf + tl) * (int)sizeof(cplx);
#pragma unroll
        for (int m = 0; m < kEPT; ++m) r[m] = bload_c(dres, voff + ((full || tl + m * Tr < in.nf_eff) ? m * kStep : 0), 0);
    };
    cplx raw[kEPT];
    double w0 = 1.0, w1 = 1.0;
    int tile = lb;
    if (tile < ntiles) load_pair(tile * SPB + i, t, raw, w0, w1);
    for (; tile < ntiles; tile += G) {
        const int p = tile * SPB + i;
        const bool active = p < a.npairs;
        const int c0 = 2 * p;
        const bool has1 = active && c0 + 1 < in.nt_eff;          // the pair's second column exists
        {
            // the frequency window at this thread's rows: sixteen unconditional loads in one batch (under the row guard they
            // were sixteen serialised round trips to the L2 per pair)
            double f[kEPT];
#pragma unroll
            for (int m = 0; m < kEPT; ++m) f[m] = 1.0;
            int tl = t; asm volatile("" : "+v"(tl));
            if (windowed) {
#pragma unroll
                for (int m = 0; m < kEPT; ++m)
                    f[m] = bload_d(fres, (tl + ((full || tl + m * Tr < in.nf) ? m * Tr : 0)) * (int)sizeof(double), 0);
                if (c0 + 1 >= in.nt) w1 = 0.0;
            }
#pragma unroll
            for (int m = 0; m < kEPT; ++m) {
                const bool ok = active && (full || t + m * Tr < in.nf_eff);
                const double d00 = sspec_d(raw[m].x, w0, f[m], m1, m2, windowed);
                const double d01 = sspec_d(raw[m].y, w1, f[m], m1, m2, windowed);
                raw[m] = mk(ok ? d00 : 0.0, (ok && has1) ? d01 : 0.0);
            }
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            cplx v[kEPT];
            if (half == 0) {
#pragma unroll
                for (int m = 0; m < kEPT; ++m) v[m] = raw[m];
            } else {
                static_for<0, kEPT>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    v[m] = mul_w32<m>(raw[m] * wt2);
                });
                // the next pair into the registers this one leaves, in flight through the second transform (unconditionally: the
                // last iteration reads its own pair again -- a conditional load is a phi, i.e. a copy and a wait right here)
                __builtin_amdgcn_sched_barrier(0);
                int tl = t; asm volatile("" : "+v"(tl));
                load_pair((tile + G < ntiles ? tile + G : tile) * SPB + i, tl, raw, w0, w1);
            }
            // (the thread index is opaque per transform: the LDS addresses are then recomputed instead of kept in registers)
            int th = t; asm volatile("" : "+v"(th));
            slot_fft_wc<R0, R1, R2, R3>(v, ldsc, th, sw);
            // natural order in LDS; thread t separates the two real spectra at m = t + k Tr < n/2 from Z[m] and its
            // partner Z[n - m] (even half) / Z[n - 1 - m] (odd half)
            lds_barrier();
#pragma unroll
            for (int q = 0; q < kEPT / LS::RL; ++q)
#pragma unroll
                for (int m = 0; m < LS::RL; ++m)
                    ldsc[lds_pad(stockham_out_index<LS::RL>(th, Tr, LS::Ns, q, m))] = v[q * LS::RL + m];
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int m = th + k * Tr, pm = half ? n - 1 - m : (n - m) & (n - 1);
                const cplx za = ldsc[lds_pad(m)], zb = ldsc[lds_pad(pm)];
                // X_c = (Z[k] + conj Z[-k]) / 2,  X_{c+1} = (Z[k] - conj Z[-k]) / (2i); Y in tiles of (4 delay rows of one
                // parity) x (one pair): 128 contiguous bytes from 4 consecutive threads
                if (active) {
                    cplx* out = a.Y + ((int64_t)(m >> 2) * a.npairs + p) * 16 + half * 8 + (m & 3) * 2;
                    gstore(out, mk(0.5 * (za.x + zb.x), 0.5 * (za.y - zb.y)));
                    gstore(out + 1, mk(0.5 * (za.y + zb.y), -0.5 * (za.x - zb.x)));
                }
            }
        }
    }
}

struct SspecRows {
    const cplx* Y; int npairs; int nt_eff;
    int nrows;                    // R/2 kept delay rows
    int C;                        // 2n
    const cplx* tw_n; const cplx* tw_2n;
    double* out;                  // [R/2][C] dB
    int prewhite; const double* pd_fd; const double* pd_td;
    int xcd_remap;
};

// 10 log10(x), series form (round 3, first version; now the path of zero, subnormal and non-finite powers and the
// check of the table form below).  The device library's log10 is 135 instructions a value --
// 4300 of this kernel's 7200 with 32 values per thread, a 115-us floor on 4096^2 by instruction issue alone.
// Here: x = 2^e m, m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s) with s = (m - 1) / (m + 1), |s| <= 0.1716,
// ten terms of the odd series (truncation 3e-16 absolute); 35 instructions, absolute error a few 1e-15 dB
// + 2e-16 |result| -- seven orders inside the 1e-8 dB parity tolerance of the path.  Zero, subnormal and
// non-finite powers are handled by selects (no branch, no library call).
__device__ __attribute__((noinline)) double ten_log10_series(double x) {
    // subnormal powers are scaled into the normal range first; 0 -> -inf, +inf and NaN pass through
    const bool tiny = x < 2.2250738585072014e-308;
    const double xn = tiny ? x * 18446744073709551616.0 : x;       // * 2^64
    const long long bits = __double_as_longlong(xn);
    int e = (int)((bits >> 52) & 0x7ff) - 1023 - (tiny ? 64 : 0);
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);   // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = r * (2.0 - den * r);
    r = r * (2.0 - den * r);
    double s = num * r;
    s = s + r * (num - den * s);                 // s = num / den to the last bit or two
    const double z = s * s;
    // 2 (1 + z/3 + z^2/5 + ... + z^9/19) * 10 / ln 10
    constexpr double k = 8.6858896380650365530225783783321;   // 20 / ln 10
    double p = k / 19.0;
    p = p * z + k / 17.0; p = p * z + k / 15.0; p = p * z + k / 13.0; p = p * z + k / 11.0;
    p = p * z + k / 9.0;  p = p * z + k / 7.0;  p = p * z + k / 5.0;  p = p * z + k / 3.0;
    p = p * z + k;
    double res = (double)e * 3.0102999566398119521373889472449 + s * p;   // e * 10 log10 2 + 10 log10 m
    if (x == 0.0) res = -INFINITY;
    if (!(x < INFINITY)) res = x;                // +inf, NaN
    return res;
}

// 10 log10(x) for the power of a spectral bin, table form: x = 2^e m, m in [1, 2); the top seven mantissa bits pick
// c = 1 + (i + 1/2) / 128 and the table holds (1/c rounded, -(10 / ln 10) ln of THAT rounded value), so
// 10 log10 m = klc + (10 / ln 10) log1p(r), r = m / c - 1 (one fma, |r| <= 2^-8), five terms of the series (the sixth is
// 6e-16).  21 instructions and no division against 38 and a reciprocal (a quarter-rate instruction) for the series
// form above -- the sixteen logarithms of a thread were 28 % of the row kernel's vector instructions
// (profiles/r03_sspec_counters.txt).  Absolute error a few 1e-15 dB + 2e-16 |result|, as before.
// tools/gen_log_table.py writes the table (60-digit arithmetic, one rounding per entry).
__device__ const double kTenLogTab[128][2] = {
    {0x1.fe01fe01fe020p-1, 0x1.156831cd1ad54p-6},
    {0x1.fa11caa01fa12p-1, 0x1.9e7fabcc74a1ep-5},
    {0x1.f6310aca0dbb5p-1, 0x1.5816114b8d2b7p-4},
    {0x1.f25f644230ab5p-1, 0x1.dfe0e3ff6205ep-4},
    {0x1.ee9c7f8458e02p-1, 0x1.33522d9f46dd4p-3},
    {0x1.eae807aba01ebp-1, 0x1.7632367028c46p-3},
    {0x1.e741aa59750e4p-1, 0x1.b8927c0493a33p-3},
    {0x1.e3a9179dc1a73p-1, 0x1.fa74e2eb771a8p-3},
    {0x1.e01e01e01e01ep-1, 0x1.1deda281b0d86p-2},
    {0x1.dca01dca01dcap-1, 0x1.3e63b8e5331afp-2},
    {0x1.d92f2231e7f8ap-1, 0x1.5e9d97558e092p-2},
    {0x1.d5cac807572b2p-1, 0x1.7e9c1ba36d410p-2},
    {0x1.d272ca3fc5b1ap-1, 0x1.9e601edea68a7p-2},
    {0x1.cf26e5c44bfc6p-1, 0x1.bdea7578bf30ap-2},
    {0x1.cbe6d9601cbe7p-1, 0x1.dd3bef663a2d1p-2},
    {0x1.c8b265afb8a42p-1, 0x1.fc55583ebc30bp-2},
    {0x1.c5894d10d4986p-1, 0x1.0d9bbbae08fc9p-1},
    {0x1.c26b5392ea01cp-1, 0x1.1cf187fc1263cp-1},
    {0x1.bf583ee868d8bp-1, 0x1.2c2c70a4f4198p-1},
    {0x1.bc4fd65883e7bp-1, 0x1.3b4cd350a60c5p-1},
    {0x1.b951e2b18ff23p-1, 0x1.4a530bc11e6a2p-1},
    {0x1.b65e2e3beee05p-1, 0x1.593f73df5dc4bp-1},
    {0x1.b37484ad806cep-1, 0x1.681263c80bdcep-1},
    {0x1.b094b31d922a4p-1, 0x1.76cc31d7a984dp-1},
    {0x1.adbe87f94905ep-1, 0x1.856d32b65bce6p-1},
    {0x1.aaf1d2f87ebfdp-1, 0x1.93f5b963548fdp-1},
    {0x1.a82e65130e159p-1, 0x1.a266173fdc15bp-1},
    {0x1.a574107688a4ap-1, 0x1.b0be9c19ffac3p-1},
    {0x1.a2c2a87c51ca0p-1, 0x1.beff9636e8839p-1},
    {0x1.a01a01a01a01ap-1, 0x1.cd29525cde489p-1},
    {0x1.9d79f176b682dp-1, 0x1.db3c1bdcf8a45p-1},
    {0x1.9ae24ea5510dap-1, 0x1.e9383c9c82b17p-1},
    {0x1.9852f0d8ec0ffp-1, 0x1.f71dfd1e13575p-1},
    {0x1.95cbb0be377aep-1, 0x1.0276d2452eaa0p+0},
    {0x1.934c67f9b2ce6p-1, 0x1.0953bc5c5cd32p+0},
    {0x1.90d4f120190d5p-1, 0x1.1025df1bb7716p+0},
    {0x1.8e6527af1373fp-1, 0x1.16ed5c29dbf7ap+0},
    {0x1.8bfce8062ff3ap-1, 0x1.1daa5490c63bfp+0},
    {0x1.899c0f601899cp-1, 0x1.245ce8c196b07p+0},
    {0x1.87427bcc092b9p-1, 0x1.2b0538983bae3p+0},
    {0x1.84f00c2780614p-1, 0x1.31a3635efedb0p+0},
    {0x1.82a4a0182a4a0p-1, 0x1.383787d1f7a74p+0},
    {0x1.8060180601806p-1, 0x1.3ec1c42263d64p+0},
    {0x1.7e225515a4f1dp-1, 0x1.454235f9e6fc9p+0},
    {0x1.7beb3922e017cp-1, 0x1.4bb8fa7db1d08p+0},
    {0x1.79baa6bb6398bp-1, 0x1.52262e5192271p+0},
    {0x1.77908119ac60dp-1, 0x1.5889ed9aec67fp+0},
    {0x1.756cac201756dp-1, 0x1.5ee454039f416p+0},
    {0x1.734f0c541fe8dp-1, 0x1.65357cbcd2575p+0},
    {0x1.713786d9c7c09p-1, 0x1.6b7d8281b0a57p+0},
    {0x1.6f26016f26017p-1, 0x1.71bc7f9a0f431p+0},
    {0x1.6d1a62681c861p-1, 0x1.77f28ddd01318p+0},
    {0x1.6b1490aa31a3dp-1, 0x1.7e1fc6b358d5ep+0},
    {0x1.691473a88d0c0p-1, 0x1.8444431a17bb0p+0},
    {0x1.6719f3601671ap-1, 0x1.8a601ba4cd300p+0},
    {0x1.6524f853b4aa3p-1, 0x1.9073687fe453bp+0},
    {0x1.63356b88ac0dep-1, 0x1.967e4172e2178p+0},
    {0x1.614b36831ae94p-1, 0x1.9c80bde293bf3p+0},
    {0x1.5f66434292dfcp-1, 0x1.a27af4d32e5c3p+0},
    {0x1.5d867c3ece2a5p-1, 0x1.a86cfcea5fc16p+0},
    {0x1.5babcc647fa91p-1, 0x1.ae56ec7151652p+0},
    {0x1.59d61f123ccaap-1, 0x1.b438d9569da4ep+0},
    {0x1.5805601580560p-1, 0x1.ba12d93037d6dp+0},
    {0x1.56397ba7c52e2p-1, 0x1.bfe5013d47958p+0},
    {0x1.54725e6bb82fep-1, 0x1.c5af6667f7aa5p+0},
    {0x1.52aff56a8054bp-1, 0x1.cb721d4738f9cp+0},
    {0x1.50f22e111c4c5p-1, 0x1.d12d3a2079d13p+0},
    {0x1.4f38f62dd4c9bp-1, 0x1.d6e0d0e951ef3p+0},
    {0x1.4d843bedc2c4cp-1, 0x1.dc8cf54923a3cp+0},
    {0x1.4bd3edda68fe1p-1, 0x1.e231ba9ab2565p+0},
    {0x1.4a27fad76014ap-1, 0x1.e7cf33edaeca8p+0},
    {0x1.4880522014880p-1, 0x1.ed657408396efp+0},
    {0x1.46dce34596066p-1, 0x1.f2f48d685b03cp+0},
    {0x1.453d9e2c776cap-1, 0x1.f87c924573e2bp+0},
    {0x1.43a2730abee4dp-1, 0x1.fdfd9491a22fap+0},
    {0x1.420b5265e5951p-1, 0x1.01bbd2fd8f9b4p+1},
    {0x1.40782d10e6566p-1, 0x1.04756bf6ca1c9p+1},
    {0x1.3ee8f42a5af07p-1, 0x1.072b9dc9b376dp+1},
    {0x1.3d5d991aa75c6p-1, 0x1.09de70eb7ef4dp+1},
    {0x1.3bd60d9232955p-1, 0x1.0c8dedb1fe9c7p+1},
    {0x1.3a524387ac822p-1, 0x1.0f3a1c543dab2p+1},
    {0x1.38d22d366088ep-1, 0x1.11e304eb17606p+1},
    {0x1.3755bd1c945eep-1, 0x1.1488af71ca31ep+1},
    {0x1.35dce5f9f2af8p-1, 0x1.172b23c687818p+1},
    {0x1.34679ace01346p-1, 0x1.19ca69aafff02p+1},
    {0x1.32f5ced6a1dfap-1, 0x1.1c6688c4ec656p+1},
    {0x1.3187758e9ebb6p-1, 0x1.1eff889e93e3cp+1},
    {0x1.301c82ac40260p-1, 0x1.219570a74e3f5p+1},
    {0x1.2eb4ea1fed14bp-1, 0x1.2428483403ce8p+1},
    {0x1.2d50a012d50a0p-1, 0x1.26b8167faa29ap+1},
    {0x1.2bef98e5a3711p-1, 0x1.2944e2abbe0d9p+1},
    {0x1.2a91c92f3c105p-1, 0x1.2bceb3c0ba762p+1},
    {0x1.293725bb804a5p-1, 0x1.2e5590ae8d03fp+1},
    {0x1.27dfa38a1ce4dp-1, 0x1.30d9804d07bf8p+1},
    {0x1.268b37cd60127p-1, 0x1.335a895c504c9p+1},
    {0x1.2539d7e9177b2p-1, 0x1.35d8b2854c9fbp+1},
    {0x1.23eb79717605bp-1, 0x1.3854025a0d45ap+1},
    {0x1.22a0122a0122ap-1, 0x1.3acc7f56354ecp+1},
    {0x1.21579804855e6p-1, 0x1.3d422fdf5fee5p+1},
    {0x1.2012012012012p-1, 0x1.3fb51a4583db6p+1},
    {0x1.1ecf43c7fb84cp-1, 0x1.422544c354853p+1},
    {0x1.1d8f5672e4abdp-1, 0x1.4492b57ea1272p+1},
    {0x1.1c522fc1ce059p-1, 0x1.46fd7288b1ccep+1},
    {0x1.1b17c67f2bae3p-1, 0x1.496581dea2512p+1},
    {0x1.19e0119e0119ep-1, 0x1.4bcae969bb687p+1},
    {0x1.18ab083902bdbp-1, 0x1.4e2daeffc9c16p+1},
    {0x1.1778a191bd684p-1, 0x1.508dd8637348ep+1},
    {0x1.1648d50fc3201p-1, 0x1.52eb6b448a9ddp+1},
    {0x1.151b9a3fdd5c9p-1, 0x1.55466d4060bfep+1},
    {0x1.13f0e8d344724p-1, 0x1.579ee3e215060p+1},
    {0x1.12c8b89edc0acp-1, 0x1.59f4d4a2e3654p+1},
    {0x1.11a3019a74826p-1, 0x1.5c4844ea71169p+1},
    {0x1.107fbbe011080p-1, 0x1.5e993a0f17a16p+1},
    {0x1.0f5edfab325a2p-1, 0x1.60e7b9562e5a0p+1},
    {0x1.0e40655826011p-1, 0x1.6333c7f452594p+1},
    {0x1.0d24456359e3ap-1, 0x1.657d6b0dacfa3p+1},
    {0x1.0c0a7868b4171p-1, 0x1.67c4a7b638e5fp+1},
    {0x1.0af2f722eecb5p-1, 0x1.6a0982f205b67p+1},
    {0x1.09ddba6af8360p-1, 0x1.6c4c01b57a391p+1},
    {0x1.08cabb37565e2p-1, 0x1.6e8c28e5955adp+1},
    {0x1.07b9f29b8eae2p-1, 0x1.70c9fd582dc48p+1},
    {0x1.06ab59c7912fbp-1, 0x1.730583d430305p+1},
    {0x1.059eea0727586p-1, 0x1.753ec111dc7ffp+1},
    {0x1.04949cc1664c5p-1, 0x1.7775b9bb019bdp+1},
    {0x1.038c6b78247fcp-1, 0x1.79aa726b38218p+1},
    {0x1.02864fc7729e9p-1, 0x1.7bdcefb01be9fp+1},
    {0x1.0182436517a37p-1, 0x1.7e0d3609846cdp+1},
    {0x1.0080402010080p-1, 0x1.803b49e9bc09dp+1},
};
// The table form without its branch: returns the table-form value for ANY bit pattern (garbage for zero, subnormal and non-finite
// powers) and ORs `special` for those; the caller recomputes the flagged ones with the series form under ONE branch per group -- a
// branch per logarithm puts every logarithm in its own scheduling region, and the sixteen of a thread then run as sixteen chains
// of dependent operations one after the other (7500 clocks per half row where the arithmetic is 1500: profiles/r06_rows2_phase_times*.txt)
__device__ inline double ten_log10_fast(double x, const double (*tab)[2], bool& special) {
    const unsigned hi = (unsigned)__double2hiint(x);
    const unsigned ef = hi >> 20;                                   // sign and exponent field
    special = special || (ef - 1u >= 2046u);
    const double m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), __double2loint(x));
    const unsigned i = (hi >> 13) & 127u;
    const double r = fma(m, tab[i][0], -1.0);
    constexpr double k = 4.3429448190325182765112891891661;         // 10 / ln 10
    double p = k / 5.0;
    p = fma(p, r, -k / 4.0); p = fma(p, r, k / 3.0); p = fma(p, r, -k / 2.0); p = fma(p, r, k);
    return fma(p, r, fma((double)((int)ef - 1023), 3.0102999566398119521373889472449, tab[i][1]));
}
__device__ inline double ten_log10(double x, const double (*tab)[2]) {
    const unsigned hi = (unsigned)__double2hiint(x);
    const unsigned ef = hi >> 20;                                   // sign and exponent field
    if (ef - 1u >= 2046u) return ten_log10_series(x);               // zero, subnormal, inf, NaN (a power is never negative)
    const double m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), __double2loint(x));
    const unsigned i = (hi >> 13) & 127u;
    const double r = fma(m, tab[i][0], -1.0);
    constexpr double k = 4.3429448190325182765112891891661;         // 10 / ln 10
    double p = k / 5.0;
    p = fma(p, r, -k / 4.0); p = fma(p, r, k / 3.0); p = fma(p, r, -k / 2.0); p = fma(p, r, k);
    return fma(p, r, fma((double)((int)ef - 1023), 3.0102999566398119521373889472449, tab[i][1]));
}

// Round 6: the row pass as a PERSISTENT, self-pipelined kernel.  What the counters of round 5 said about the kernel above
// (profiles/r05_sspec_roofline_4096.json): its HBM floor is 93 us, its issue floor 81 us, it takes 222 -- the sum of its phases:
// single-shot workgroups load, transform and store one after the other, and because every phase is bound by a shared resource the
// workgroups of the chip fall into step (all load, then all compute).  Here a workgroup walks over rows and does BOTH halves of a
// row from ONE read of it:
//   * the 16 values a thread loaded stay in registers through the even half's transform and are the odd half's input too (times
//     W_2n^s = W_2n^t W_32^m: one table entry per THREAD for the whole kernel, the rest are the radix-32 constants) -- the row
//     is read once, not twice;
//   * as soon as the odd half has taken its input, the same registers receive the NEXT row: its loads are in flight through the
//     whole second transform (the transform's barriers wait for the LDS counter only, see xbarrier);
//   * a thread ends up holding bins 2 mm (even half) and 2 mm + 1 (odd half) of the same mm: one 16-byte store per pair, consecutive
//     lanes 16 bytes apart -- whole 128-byte lines (the single-half workgroups stored 8 bytes every 16 and left the merging to the
//     L2: 315 MB written for a 268 MB output).
// Workgroups per CU are set by registers (two waves per SIMD); logical blocks are dealt to XCDs in contiguous runs so that the four
// rows of one parity that share the lines of a tile row meet in one L2.
template <int R0, int R1, int R2, int R3>
__global__ void __launch_bounds__((R0 * R1 * R2 * R3 / kEPT) >= 256 ? (R0 * R1 * R2 * R3 / kEPT) : 256, 2)
sspec_rows2_kernel(SspecRows a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    static_assert(R0 == kEPT, "one radix-16 butterfly per thread in stage 0: input s = t + m n/16");
    using LS = LastStage<R0, R1, R2, R3>;
    constexpr int kLG = SCINT_ROWS2_LOG_GROUP;                      // logarithms per scheduling group
    const int i = (int)threadIdx.x / Tr, t = (int)threadIdx.x - i * Tr;
    const int spb = (int)blockDim.x / Tr, G = (int)gridDim.x;
    int lb = (int)blockIdx.x;
    if (a.xcd_remap) lb = (lb & 7) * (G >> 3) + (lb >> 3);
    double* lds = reinterpret_cast<double*>(smem_raw) + (size_t)i * lds_pad(n);
    // the even half's sixteen bins of a thread wait in LDS for the odd half's (thread-private slots, lanes contiguous: no barrier,
    // no conflict) -- in registers they cost the kernel its prefetch (spills at two waves per SIMD)
    double* evl = reinterpret_cast<double*>(smem_raw) + (size_t)spb * lds_pad(n) + threadIdx.x;
    const int evs = (int)blockDim.x;
    __shared__ double tlog[128][2];                                 // ten_log10's table (visible after the transform's barriers)
    if (threadIdx.x < 128) { tlog[threadIdx.x][0] = kTenLogTab[threadIdx.x][0]; tlog[threadIdx.x][1] = kTenLogTab[threadIdx.x][1]; }
    const cplx wt = a.tw_2n[t];                                     // W_2n^t, t < n/16
    StageW<R0, R1, R2, R3> sw;
    sw.load(t, a.tw_n);
    const int ngroups = (a.nrows + spb - 1) / spb;
    // row k1 = 2 m + h of the tiled intermediate: element c at tile (m >> 2, c >> 1), slot h, m & 3, c & 1.  Every load is
    // unconditional (a load under a branch makes the compiler wait for everything in flight) and goes through a buffer resource:
    // ONE per-lane byte offset (row + thread) and sixteen scalar distances -- as global loads the compiler kept sixteen
    // loop-carried 64-bit addresses, and spilled them.  Elements beyond the row's length read element t and are zeroed by selects.
    const bool full = a.nt_eff == n;
    const BufRsrc yres = make_rsrc(a.Y, (int64_t)a.nrows * a.npairs * 2 * (int64_t)sizeof(cplx));
    constexpr int kStep = (Tr / 2) * 16 * (int)sizeof(cplx);           // from s to s + n/16: n/32 tiles on
    // (no select and no branch between the loads and their first use in the NEXT iteration: either makes the compiler wait for
    //  the data where it is issued.  The zeroing of a short row's tail happens where the values are consumed.)
    auto row_voff = [&](int k1, int tl) {
        const int kk = k1 < a.nrows ? k1 : 0;
        return (((kk >> 3) * a.npairs + (tl >> 1)) * 16 + (kk & 1) * 8 + ((kk >> 1) & 3) * 2 + (tl & 1)) * (int)sizeof(cplx);
    };
    constexpr int step = kStep;
    // elements m with m * NP / 16 == P: piece P of NP of a row's loads
    auto load_piece = [&](auto pc, auto npc, int voff, int tl, cplx (&r)[kEPT]) {
        constexpr int P = decltype(pc)::value, NP = decltype(npc)::value;
#pragma unroll
        for (int m = 0; m < kEPT; ++m)
            if (m * NP / kEPT == P) r[m] = bload_c(yres, voff + ((full || tl + m * Tr < a.nt_eff) ? m * step : 0), 0);
    };
    auto load_row = [&](int k1, int tl, cplx (&r)[kEPT]) {
        load_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, row_voff(k1, tl), tl, r);
    };
    // A row's bins are stored at the top of the NEXT iteration (the odd half's sixteen values ride in registers across the loop
    // edge, where nothing else is alive; the even half's wait in LDS anyway): the wait for the prefetched row is then a wait for
    // loads only.  Stored where they are computed, the sixteen stores sit behind the prefetch on the in-order counter and the
    // compiler's one static wait at the loop head drains them too -- a store round trip per row, fully exposed.
    double od[kEPT];
    int pk = -1;
    // bins e with e * NP / 16 == P of the previous row
    auto flush_piece = [&](auto pc, auto npc) {
        constexpr int P = decltype(pc)::value, NP = decltype(npc)::value;
        if (pk >= 0) {
            int tq = t; asm volatile("" : "+v"(tq));
            double* __restrict__ orow = a.out + (int64_t)pk * a.C;
#pragma unroll
            for (int q = 0; q < kEPT / LS::RL; ++q) {
#pragma unroll
                for (int m = 0; m < LS::RL; ++m) {
                    const int e = q * LS::RL + m;
                    if (e * NP / kEPT != P) continue;
                    const int col = (2 * stockham_out_index<LS::RL>(tq, Tr, LS::Ns, q, m) + n) & (a.C - 1);
                    v2d o; o.x = evl[e * evs]; o.y = od[e];
                    *(SCINT_GLOBAL v2d*)(orow + col) = o;
                }
            }
        }
    };
    auto flush = [&]() { flush_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); };
    cplx raw[kEPT];
    int grp = lb;
    if (grp < ngroups) load_row(grp * spb + i, t, raw);
    for (; grp < ngroups; grp += G) {
        const int k1 = grp * spb + i;
        const bool active = k1 < a.nrows;
        const double td = (a.prewhite && active) ? a.pd_td[k1] : 1.0;
        if (!full || !active) {
#pragma unroll
            for (int m = 0; m < kEPT; ++m) {
                const bool ok = active && t + m * Tr < a.nt_eff;
                raw[m] = mk(ok ? raw[m].x : 0.0, ok ? raw[m].y : 0.0);
            }
        }
        cplx v[kEPT];
#pragma unroll
        for (int m = 0; m < kEPT; ++m) {
            // (a use of every loaded value HERE: the wait for the row must precede the stores below, or it becomes a wait for them)
            double rx = raw[m].x, ry = raw[m].y;
            asm volatile("" : "+v"(rx));
            asm volatile("" : "+v"(ry));
            v[m] = mk(rx, ry);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the thread index of each transform is opaque to the compiler: it would otherwise keep every LDS address of the exchanges
        // alive across both transforms and the whole loop (hundreds of registers, see the note above sspec_rows_kernel)
        int th = t; asm volatile("" : "+v"(th));
        // the previous row's bins leave in kSP pieces, before the first stage and after the first kSP - 1 stages
        constexpr int kSP = SCINT_ROWS2_STORE_PIECES;
        slot_fft_w<R0, R1, R2, R3>(v, lds, th, sw, [&](auto c) {
            if constexpr (decltype(c)::value < kSP) {
                __builtin_amdgcn_sched_barrier(0);
                flush_piece(c, std::integral_constant<int, kSP>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        static_for<0, kEPT / kLG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            double pw[kLG], lg[kLG];
            bool special = false;
#pragma unroll
            for (int j = 0; j < kLG; ++j) {
                constexpr int RL = LS::RL;
                const int e = g * kLG + j, q = e / RL, m = e % RL;
                const int mm = stockham_out_index<RL>(th, Tr, LS::Ns, q, m);
                const int col = (2 * mm + n) & (a.C - 1);           // Doppler bin 2 mm at its fftshift-ed place (dynspec.py:3687)
                double p = v[e].x * v[e].x + v[e].y * v[e].y;
                if (a.prewhite) {   // post-darkening, column C/2 and row 0 forced to 1 (dynspec.py:3704-3717)
                    const double d = (col == n || k1 == 0) ? 1.0 : a.pd_fd[col] * td;
                    p = p / d;
                }
                pw[j] = p;
                lg[j] = ten_log10_fast(p, tlog, special);
            }
            if (special) {                                          // zero, subnormal, infinite or NaN power somewhere in the group
#pragma unroll
                for (int j = 0; j < kLG; ++j) lg[j] = ten_log10(pw[j], tlog);
            }
#pragma unroll
            for (int j = 0; j < kLG; ++j) evl[(g * kLG + j) * evs] = lg[j];
            __builtin_amdgcn_sched_barrier(0);                      // a group of logarithms at a time (all sixteen at once spill)
        });
        // the odd half: the row times W_2n^s, s = t + m n/16
        static_for<0, kEPT>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            v[m] = mul_w32<m>(raw[m] * wt);
        });
        // ... and the next row into the registers this one leaves (unconditionally: the last iteration reads its own row again --
        // a conditional load is a phi, the compiler then loads elsewhere and copies, i.e. waits, right here)
        __builtin_amdgcn_sched_barrier(0);
        th = t; asm volatile("" : "+v"(th));
        constexpr int kLP = SCINT_ROWS2_LOAD_PIECES < HookCount<R0, R1, R2, R3>::value ? SCINT_ROWS2_LOAD_PIECES : HookCount<R0, R1, R2, R3>::value - 1;
        const int nvoff = row_voff((grp + G < ngroups ? grp + G : grp) * spb + i, th);
        slot_fft_w<R0, R1, R2, R3>(v, lds, th, sw, [&](auto c) {
            if constexpr (decltype(c)::value < kLP) {
                __builtin_amdgcn_sched_barrier(0);
                load_piece(c, std::integral_constant<int, kLP>{}, nvoff, th, raw);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        static_for<0, kEPT / kLG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            double pw[kLG], lg[kLG];
            bool special = false;
#pragma unroll
            for (int j = 0; j < kLG; ++j) {
                constexpr int RL = LS::RL;
                const int e = g * kLG + j, q = e / RL, m = e % RL;
                const int mm = stockham_out_index<RL>(th, Tr, LS::Ns, q, m);
                const int col = (2 * mm + n) & (a.C - 1);
                double p = v[e].x * v[e].x + v[e].y * v[e].y;
                if (a.prewhite) {
                    const double d = (k1 == 0) ? 1.0 : a.pd_fd[col + 1] * td;    // (col + 1 is odd: never C/2)
                    p = p / d;
                }
                pw[j] = p;
                lg[j] = ten_log10_fast(p, tlog, special);
            }
            if (special) {
#pragma unroll
                for (int j = 0; j < kLG; ++j) lg[j] = ten_log10(pw[j], tlog);
            }
#pragma unroll
            for (int j = 0; j < kLG; ++j) od[g * kLG + j] = lg[j];
            __builtin_amdgcn_sched_barrier(0);
        });
        pk = active ? k1 : -1;
    }
    flush();
}

template <int R0, int R1, int R2, int R3>
static int32_t launch_cols(const SspecCols& a, hipStream_t stream) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT, kColsBlock = ColsBlock<n>::value, SPB = kColsBlock / Tr;
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

// SCINT_SSPEC_MAXGRID (tests; read per call): an upper bound on the workgroups of the persistent kernels, so that small
// shapes run several iterations per workgroup on the host interpreter of the kernels (tests/test_emu_cpu.py)
static int sspec_max_grid() {
    const char* e = getenv("SCINT_SSPEC_MAXGRID");
    return e ? atoi(e) : 0;
}
constexpr int kSspecCUs = 256;     // compute units of an MI355X: the persistent kernels launch (workgroups per CU) x this many
static int persistent_grid(int units, int per_cu) {
    int g = per_cu * kSspecCUs;
    if (sspec_max_grid() > 0 && g > sspec_max_grid()) g = sspec_max_grid();
    if (g > units) g = units;
    if (g >= 8) g &= ~7;           // whole XCD rounds (the logical-block remap)
    return g < 1 ? 1 : g;
}

template <int R0, int R1, int R2, int R3>
static int32_t launch_cols2(const SspecCols& a, hipStream_t stream) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT, kColsBlock = ColsBlock<n>::value, SPB = kColsBlock / Tr;
    const int tiles = (int)ceil_div(a.npairs, SPB);
    const int per_cu = kColsBlock >= 512 ? 1 : 2;      // two waves per SIMD (registers)
    const int grid = persistent_grid(tiles, per_cu);
    SspecCols b = a;
    b.xcd_remap = (grid % 8 == 0) ? 1 : 0;
    const size_t lds = (size_t)SPB * (size_t)(n + n / 16) * sizeof(cplx);      // whole complex values (68 KiB at n = 4096: two per CU)
    auto k = sspec_cols2_kernel<R0, R1, R2, R3>;
    SCINT_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kColsBlock), lds, stream, b);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

template <int R0, int R1, int R2, int R3>
static int32_t launch_rows2(const SspecRows& a, hipStream_t stream) {
    constexpr int n = R0 * R1 * R2 * R3, Tr = n / kEPT;
    const int block = Tr >= 256 ? Tr : 256, spb = block / Tr;
    const int groups = (int)ceil_div((int64_t)a.nrows, spb);
    const int per_cu = block >= 512 ? 1 : 2;           // two waves per SIMD (registers)
    const int grid = persistent_grid(groups, per_cu);
    SspecRows b = a;
    b.xcd_remap = (grid % 8 == 0) ? 1 : 0;
    const size_t lds = ((size_t)spb * (size_t)(n + n / 16) + (size_t)kEPT * (size_t)block) * sizeof(double);
    auto k = sspec_rows2_kernel<R0, R1, R2, R3>;
    if (lds > 64 * 1024)
        SCINT_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, b);
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
    if (!a.in.prewhite) { SCINT_SSPEC_DISPATCH(n, launch_cols2, a, s) }
    SCINT_SSPEC_DISPATCH(n, launch_cols, a, s)
    SCINT_REQUIRE(false, "sspec: unsupported column transform length");
}
static int32_t dispatch_rows(int64_t n, const SspecRows& a, hipStream_t s) {
    SCINT_SSPEC_DISPATCH(n, launch_rows2, a, s)
    SCINT_REQUIRE(false, "sspec: unsupported row transform length");
}

bool sspec_fast_supported(int64_t nf, int64_t nt, int32_t halve) {
    if (!halve || nf < 3 || nt < 3) return false;
    const int64_t nr = next_pow2(nf), nc = next_pow2(nt);   // R/2, C/2
    return nr >= 256 && nr <= 8192 && nc >= 256 && nc <= 8192;
}

struct SspecWs { size_t Y, dynp, partial, scal, total; int tiles_x, tiles_y; };
static SspecWs sspec_ws(int64_t nf, int64_t nt) {
    SspecWs w;
    const int64_t nr = next_pow2(nf), ntp = 2 * ceil_div(nt, 2);
    w.tiles_x = (int)ceil_div(nt, 64); w.tiles_y = (int)ceil_div(nf, 64);
    size_t off = 0;
    auto take = [&](size_t bytes) { off = align_up(off, 256); const size_t o = off; off += bytes; return o; };
    w.Y = take(sizeof(cplx) * (size_t)nr * (size_t)ntp);
    w.dynp = take(sizeof(double) * (size_t)ntp * (size_t)nf);
    w.partial = take(sizeof(double) * 3 * (size_t)w.tiles_x * (size_t)w.tiles_y);
    w.scal = take(sizeof(double) * 8);
    w.total = align_up(off, 256);
    return w;
}
size_t sspec_fast_workspace(int64_t nf, int64_t nt) { return sspec_ws(nf, nt).total; }

int32_t sspec_fast(const double* dyn, int64_t nf, int64_t nt, const double* win_t, const double* win_f,
                   int32_t prewhite, const double* pd_fd, const double* pd_td,
                   double* sec_out, void* workspace, hipStream_t stream) {
    const int64_t nr = next_pow2(nf), nc = next_pow2(nt);
    const int64_t nf_eff = prewhite ? nf - 1 : nf, nt_eff = prewhite ? nt - 1 : nt;
    const cplx* tw_r = twiddle_table(nr);
    const cplx* tw_2r = twiddle_table(2 * nr);
    const cplx* tw_c = twiddle_table(nc);
    const cplx* tw_2c = twiddle_table(2 * nc);
    if (!tw_r || !tw_2r || !tw_c || !tw_2c) return SCINT_E_HIP;
    const SspecWs ws = sspec_ws(nf, nt);
    char* base = (char*)workspace;
    double* dynp = (double*)(base + ws.dynp);
    double* partial = (double*)(base + ws.partial);
    double* scal = (double*)(base + ws.scal);
    const int p0 = profiler().begin(kProfSspecPrep, stream);
    hipLaunchKernelGGL(sspec_prep_kernel, dim3((unsigned)ws.tiles_x, (unsigned)ws.tiles_y), dim3(256), 0, stream, dyn, win_t,
                       win_f, (int)nf, (int)nt, dynp, partial);
    const bool cols_persistent = !prewhite;     // (the prewhitening stencil keeps round 5's column kernel: three more loads per value, not prefetched)
    if (!cols_persistent)
        hipLaunchKernelGGL(sspec_prep_means_kernel, dim3(1), dim3(256), 0, stream, partial, ws.tiles_x * ws.tiles_y,
                           (double)(nf * nt), scal);
    profiler().end(kProfSspecPrep, p0, stream);
    SCINT_LAUNCH_CHECK();
    SspecCols ca{};
    ca.in = SspecIn{dynp, win_t, win_f, scal, (int)nf, (int)nt, (int)nf_eff, (int)nt_eff, prewhite, partial, ws.tiles_x * ws.tiles_y};
    ca.npairs = (int)ceil_div(nt_eff, 2);
    ca.Y = (cplx*)(base + ws.Y); ca.ldY = 2 * ca.npairs;
    ca.tw_n = tw_r; ca.tw_2n = tw_2r;
    const int p1 = profiler().begin(kProfSspecCols, stream);
    int32_t rc = dispatch_cols(nr, ca, stream);
    profiler().end(kProfSspecCols, p1, stream);
    if (rc != SCINT_OK) return rc;
    SspecRows ra{};
    ra.Y = ca.Y; ra.npairs = ca.npairs; ra.nt_eff = (int)nt_eff; ra.nrows = (int)nr; ra.C = (int)(2 * nc);
    ra.tw_n = tw_c; ra.tw_2n = tw_2c; ra.out = sec_out;
    ra.prewhite = prewhite; ra.pd_fd = pd_fd; ra.pd_td = pd_td;
    const int p2 = profiler().begin(kProfSspecRows, stream);
    rc = dispatch_rows(nc, ra, stream);
    profiler().end(kProfSspecRows, p2, stream);
    return rc;
}

}  // namespace scint
