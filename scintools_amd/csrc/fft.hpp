// fft.hpp -- hand-written fp64 complex FFT building blocks for gfx950.
//
// Two kernels carry every transform in the library:
//
//   fft_rows_kernel   FFT along the CONTIGUOUS axis.  One "slot" (a row, or one
//                     decimated sub-sequence of a long row) lives in LDS; every
//                     thread keeps 16 points in registers, Stockham radix-16/8/4/2
//                     stages exchange through a padded (bank-conflict-free) LDS
//                     buffer.  The first stage loads straight from global through a
//                     Loader functor (this is where mean-subtract / window /
//                     prewhiten / zero-pad are fused), the last stage stores
//                     straight to global through a Storer functor.  Both expose
//                     open(slot) -> per-slot accessor, so the slot is decoded once per
//                     thread and an element access costs one address.
//
//   fft_cols_kernel   one decimation-in-frequency radix-R pass along the STRIDED
//                     axis.  Lanes run along the contiguous axis, so every load and
//                     store is a coalesced 16 B/lane access whatever the stride;
//                     each thread holds one radix-R butterfly in registers.  Passes
//                     are in place; the LAST pass un-scrambles the digit-reversed
//                     order while storing through a Storer functor (this is where
//                     fftshift, |.|^2, post-darkening and 10 log10 are fused).
//
// All transforms are forward (numpy sign convention, exp(-2 pi i jk/n)); inverse
// transforms are expressed as conj-forward-conj by the functors.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"

namespace scint {

// Device table W_n^j = exp(-2 pi i j / n), j = 0..n-1, computed in long double on
// the host once per n and cached for the life of the process (mutex-guarded).
// Returns nullptr (and sets the error text) on allocation failure.
const cplx* twiddle_table(int64_t n);

// ------------------------------------------------------------------------------
// in-register small FFTs (natural-order output)
// ------------------------------------------------------------------------------
__device__ constexpr double kC32[9] = {
    1.0,
    0.98078528040323044913,  // cos(2 pi 1/32)
    0.92387953251128675613,  // cos(2 pi 2/32)
    0.83146961230254523708,
    0.70710678118654752440,
    0.55557023301960222474,
    0.38268343236508977173,
    0.19509032201612826785,
    0.0};

// multiply by W_32^J = exp(-2 pi i J/32), 0 <= J < 16, with the trivial cases folded
template <int J>
__device__ inline cplx mul_w32(cplx a) {
    static_assert(J >= 0 && J < 16, "J range");
    if constexpr (J == 0) {
        return a;
    } else if constexpr (J == 8) {
        return mul_mi(a);
    } else if constexpr (J == 4) {
        return mk((a.x + a.y) * kC32[4], (a.y - a.x) * kC32[4]);
    } else if constexpr (J == 12) {
        return mk((a.y - a.x) * kC32[4], -(a.x + a.y) * kC32[4]);
    } else {
        constexpr double c = (J <= 8) ? kC32[J] : -kC32[16 - J];
        constexpr double s = (J <= 8) ? kC32[8 - J] : kC32[J - 8];
        // (x + iy)(c - is)
        return mk(a.x * c + a.y * s, a.y * c - a.x * s);
    }
}

template <int N>
struct SmallFFT;

template <>
struct SmallFFT<1> {
    __device__ static inline void run(cplx (&)[1]) {}
};
template <>
struct SmallFFT<2> {
    __device__ static inline void run(cplx (&x)[2]) {
        cplx a = x[0], b = x[1];
        x[0] = a + b;
        x[1] = a - b;
    }
};
template <>
struct SmallFFT<4> {
    __device__ static inline void run(cplx (&x)[4]) {
        cplx t0 = x[0] + x[2], t1 = x[0] - x[2], t2 = x[1] + x[3], t3 = mul_mi(x[1] - x[3]);
        x[0] = t0 + t2;
        x[1] = t1 + t3;
        x[2] = t0 - t2;
        x[3] = t1 - t3;
    }
};

template <int N, int M>
struct HalfStep {
    // a[m] = x[m] + x[m+N/2]; b[m] = (x[m] - x[m+N/2]) W_N^m   for m = M..N/2-1
    __device__ static inline void run(cplx (&x)[N], cplx (&a)[N / 2], cplx (&b)[N / 2]) {
        if constexpr (M < N / 2) {
            a[M] = x[M] + x[M + N / 2];
            b[M] = mul_w32<M*(32 / N)>(x[M] - x[M + N / 2]);
            HalfStep<N, M + 1>::run(x, a, b);
        }
    }
};

template <int N>
struct SmallFFT {
    static_assert(N == 8 || N == 16 || N == 32, "radix");
    __device__ static inline void run(cplx (&x)[N]) {
        cplx a[N / 2], b[N / 2];
        HalfStep<N, 0>::run(x, a, b);
        SmallFFT<N / 2>::run(a);
        SmallFFT<N / 2>::run(b);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            x[2 * k] = a[k];
            x[2 * k + 1] = b[k];
        }
    }
};

// ------------------------------------------------------------------------------
// rows: FFT along the contiguous axis, one slot per n/16 threads
// ------------------------------------------------------------------------------
constexpr int kEPT = 16;  // points per thread

__device__ inline int lds_pad(int i) { return i + (i >> 4); }  // +1 cplx per 16

struct RowShape {
    int n;            // FFT length (power of two, 16..8192)
    int log2n;
    int threads_per_slot;  // n/16
    int slots_per_block;
    int64_t nslots;   // total slots in the launch
    const cplx* tw;   // W_n table
};

// One Stockham stage on the registers of a thread.
//   v[q*R + m], q < 16/R : inputs  in[j_q + m*(n/R)],   j_q = t + q*Tr
//   outputs out[(j_q/Ns)*Ns*R + (j_q%Ns) + m*Ns]  returned in the same slots.
template <int R>
__device__ inline void stockham_compute(cplx (&v)[kEPT], int t, int Tr, int n, int Ns,
                                        const cplx* __restrict__ tw) {
    constexpr int NB = kEPT / R;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int j = t + q * Tr;
        cplx x[R];
#pragma unroll
        for (int m = 0; m < R; ++m) x[m] = v[q * R + m];
        if (Ns > 1) {
            const int k = j & (Ns - 1);
            const int step = k * (n / (Ns * R));  // index into W_n of W_{Ns R}^k
            // x[m] *= W^(step m), m = 1 .. R-1.  ONE table read (W^step: consecutive lanes read consecutive or
            // 16-fold repeated entries) and the powers by multiplication, two interleaved chains of depth R/2
            // (relative error of the last power ~ R/2 roundings: 1e-15, five orders inside the tightest
            // tolerance of any transform here).  Reading each power from the table costs R-1 loads whose lanes are
            // step*m entries apart -- in the last stage of a 4096-point row up to 64 cache lines per wave
            // instruction, ~600 line accesses per wave and stage against 128 for the row's own data; the row
            // kernels then sit at 39 % VALU-busy waiting for the texture path (profiles/r03_sspec_counters.txt).
            if constexpr (R >= 4) {
                const cplx w1 = tw[step];                      // step < n / R
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
#pragma unroll
                for (int m = 1; m < R; ++m) x[m] = x[m] * tw[(step * m) & (n - 1)];
            }
        }
        SmallFFT<R>::run(x);
#pragma unroll
        for (int m = 0; m < R; ++m) v[q * R + m] = x[m];
    }
}

template <int R>
__device__ inline int stockham_out_index(int t, int Tr, int Ns, int q, int m) {
    const int j = t + q * Tr;
    return (j / Ns) * Ns * R + (j & (Ns - 1)) + m * Ns;
}

// A storer may REDUCE instead of (or besides) storing: it declares `static constexpr bool kReduce = true`, its
// slot accessor takes a running sum -- sx(index, value, acc) -- and the kernel leaves the workgroup's total
// (fixed order: lanes, then waves) in st.partial[blockIdx.x]; the caller adds the partials in order.
template <class S, class = void> struct IsReducing : std::false_type {};
template <class S> struct IsReducing<S, std::void_t<decltype(S::kReduce)>> : std::true_type {};

// Stage sequence R0,R1,R2,R3 (1 = unused); product = n.
// SPLIT: the stage exchanges move the real parts, then the imaginary parts, through an LDS buffer of
// n doubles per slot instead of n complex values.  These kernels' occupancy is set by LDS (a slot
// needs its whole transform there), so halving it doubles the waves per CU -- worth two more
// barriers per exchange for the short transforms of the tiled column passes, which are pure
// streaming (sspec 4096^2: 0.81 -> 0.76 ms).  Measured with every length split (incl. the pair
// storers): no gain for the long row transforms, so only lengths 32..128 are instantiated split.
template <int R0, int R1, int R2, int R3, bool SPLIT, class Loader, class Storer>
__global__ void __launch_bounds__(512)
fft_rows_kernel(RowShape sh, Loader ld, Storer st) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* smem = reinterpret_cast<cplx*>(smem_raw);
    const int n = sh.n, Tr = sh.threads_per_slot;
    const int slot_in_block = threadIdx.x / Tr;
    const int t = threadIdx.x - slot_in_block * Tr;
    const int64_t slot = (int64_t)blockIdx.x * sh.slots_per_block + slot_in_block;
    const bool active = slot < sh.nslots;
    cplx* lds = smem + (SPLIT ? 0 : (size_t)slot_in_block * lds_pad(n));
    double* ldsd = reinterpret_cast<double*>(smem_raw) + (size_t)slot_in_block * lds_pad(n);
    static_assert(!(SPLIT && Storer::kPair), "split exchange: not with pair storers");
    const cplx* __restrict__ tw = sh.tw;

    cplx v[kEPT];
    // ---- stage 0: inputs straight from global ---------------------------------
    // open(slot) decodes the slot ONCE (tile, column, row base ...); the per-element call then costs
    // an address and the access.  (Left to the compiler, the decode was repeated in each of the 16
    // unrolled element accesses.)
    const auto lx = ld.open(slot);
    {
        constexpr int NB = kEPT / R0;
        if (active) {
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int m = 0; m < R0; ++m) v[q * R0 + m] = lx(t + q * Tr + m * (n / R0));
        } else {
#pragma unroll
            for (int e = 0; e < kEPT; ++e) v[e] = mk(0.0, 0.0);
        }
    }
    int Ns = 1;
    stockham_compute<R0>(v, t, Tr, n, Ns, tw);

    auto exchange = [&](auto rprev_tag, auto rnext_tag) {
        constexpr int RP = decltype(rprev_tag)::value;
        constexpr int RN = decltype(rnext_tag)::value;
        if constexpr (SPLIT) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
                for (int m = 0; m < RP; ++m)
                    ldsd[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m].x;
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
                for (int m = 0; m < RP; ++m)
                    ldsd[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m].y;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
                for (int m = 0; m < RN; ++m)
                    v[q * RN + m] = mk(re[q * RN + m], ldsd[lds_pad(t + q * Tr + m * (n / RN))]);
            return;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kEPT / RP; ++q)
#pragma unroll
            for (int m = 0; m < RP; ++m)
                lds[lds_pad(stockham_out_index<RP>(t, Tr, Ns, q, m))] = v[q * RP + m];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kEPT / RN; ++q)
#pragma unroll
            for (int m = 0; m < RN; ++m) v[q * RN + m] = lds[lds_pad(t + q * Tr + m * (n / RN))];
    };

    if constexpr (R1 > 1) {
        exchange(std::integral_constant<int, R0>{}, std::integral_constant<int, R1>{});
        Ns *= R0;
        stockham_compute<R1>(v, t, Tr, n, Ns, tw);
        if constexpr (R2 > 1) {
            exchange(std::integral_constant<int, R1>{}, std::integral_constant<int, R2>{});
            Ns *= R1;
            stockham_compute<R2>(v, t, Tr, n, Ns, tw);
            if constexpr (R3 > 1) {
                exchange(std::integral_constant<int, R2>{}, std::integral_constant<int, R3>{});
                Ns *= R2;
                stockham_compute<R3>(v, t, Tr, n, Ns, tw);
            }
        }
    }
    // ---- last stage: outputs straight to global ---------------------------------
    constexpr int RL = (R3 > 1) ? R3 : (R2 > 1) ? R2 : (R1 > 1) ? R1 : R0;
    const auto sx = st.open(slot);
    if constexpr (Storer::kPair) {
        // Two real rows were transformed as one complex sequence z = x1 + i x2.  Put Z back
        // into LDS in natural order and let the storer separate X1[k], X2[k] from Z[k] and
        // Z[n-k] for k = 0 .. n/2 (the other half follows from conjugate symmetry).
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kEPT / RL; ++q)
#pragma unroll
            for (int m = 0; m < RL; ++m)
                lds[lds_pad(stockham_out_index<RL>(t, Tr, Ns, q, m))] = v[q * RL + m];
        __syncthreads();
        if (active) {
            for (int k = t; k <= n / 2; k += Tr)
                sx.pair(k, lds[lds_pad(k)], lds[lds_pad((n - k) & (n - 1))]);
        }
    } else if constexpr (IsReducing<Storer>::value) {
        __shared__ double red_[16];
        double acc = 0.0;
        if (active) {
#pragma unroll
            for (int q = 0; q < kEPT / RL; ++q)
#pragma unroll
                for (int m = 0; m < RL; ++m)
                    sx(stockham_out_index<RL>(t, Tr, Ns, q, m), v[q * RL + m], acc);
        }
        acc = block_sum(acc, red_);
        if (threadIdx.x == 0) st.partial[blockIdx.x] = acc;
    } else {
        if (active) {
#pragma unroll
            for (int q = 0; q < kEPT / RL; ++q)
#pragma unroll
                for (int m = 0; m < RL; ++m)
                    sx(stockham_out_index<RL>(t, Tr, Ns, q, m), v[q * RL + m]);
        }
    }
}

// Host-side launcher: picks the stage sequence for n and launches.  LO..HI bounds log2(n) at compile
// time: only those kernels are instantiated for the (Loader, Storer) pair (the tiled column passes
// use 16..128 points, the decimated long rows 4096).
template <int LO = 4, int HI = 13, class Loader, class Storer>
int32_t launch_fft_rows(int64_t n, int64_t nslots, Loader ld, Storer st, hipStream_t stream) {
    SCINT_REQUIRE(is_pow2(n) && n >= 16 && n <= 8192, "fft rows: n must be a power of two in [16, 8192]");
    if (nslots <= 0) return SCINT_OK;
    const cplx* tw = twiddle_table(n);
    if (!tw) return SCINT_E_HIP;
    RowShape sh;
    sh.n = (int)n;
    sh.log2n = ilog2(n);
    sh.threads_per_slot = (int)(n / kEPT);
    // short transforms (the 16..128-point steps of the tiled column passes, short rows) run in
    // 128-thread workgroups (17 KiB of LDS with the split exchange), longer ones in >= 256 threads
    const int min_block = n <= 128 ? 128 : 256;
    const int block = sh.threads_per_slot >= min_block ? sh.threads_per_slot : min_block;
    sh.slots_per_block = block / sh.threads_per_slot;
    sh.nslots = nslots;
    sh.tw = tw;
    const int64_t grid = ceil_div(nslots, sh.slots_per_block);
    const bool split = n >= 32 && n <= 128 && !Storer::kPair;
    const size_t lds = (size_t)sh.slots_per_block * (size_t)(n + n / 16) * (split ? sizeof(double) : sizeof(cplx));
#define SCINT_ROWS_K(R0, R1, R2, R3, SP)                                                        \
    do {                                                                                        \
        auto k = fft_rows_kernel<R0, R1, R2, R3, SP, Loader, Storer>;                           \
        if (lds > 64 * 1024)                                                                    \
            SCINT_HIP(hipFuncSetAttribute((const void*)k,                                       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, sh, ld, st);      \
    } while (0)
#define SCINT_ROWS(R0, R1, R2, R3) SCINT_ROWS_K(R0, R1, R2, R3, false)
#define SCINT_ROWS_SHORT(R0, R1, R2, R3)                                                        \
    do {                                                                                        \
        if constexpr (!Storer::kPair) {                                                         \
            if (split) { SCINT_ROWS_K(R0, R1, R2, R3, true); break; }                           \
        }                                                                                       \
        SCINT_ROWS_K(R0, R1, R2, R3, false);                                                    \
    } while (0)
#define SCINT_CASE(L, MAC, R0, R1, R2, R3)                                                      \
    case L:                                                                                     \
        if constexpr (LO <= L && L <= HI) { MAC(R0, R1, R2, R3); }                               \
        else { SCINT_REQUIRE(false, "fft rows: length outside the instantiated range"); }       \
        break;
    switch (sh.log2n) {
        SCINT_CASE(4, SCINT_ROWS, 16, 1, 1, 1)
        SCINT_CASE(5, SCINT_ROWS_SHORT, 16, 2, 1, 1)
        SCINT_CASE(6, SCINT_ROWS_SHORT, 16, 4, 1, 1)
        SCINT_CASE(7, SCINT_ROWS_SHORT, 16, 8, 1, 1)
        SCINT_CASE(8, SCINT_ROWS, 16, 16, 1, 1)
        SCINT_CASE(9, SCINT_ROWS, 16, 16, 2, 1)
        SCINT_CASE(10, SCINT_ROWS, 16, 16, 4, 1)
        SCINT_CASE(11, SCINT_ROWS, 16, 16, 8, 1)
        SCINT_CASE(12, SCINT_ROWS, 16, 16, 16, 1)
        SCINT_CASE(13, SCINT_ROWS, 16, 16, 16, 2)
        default: SCINT_REQUIRE(false, "fft rows: unsupported length");
    }
#undef SCINT_CASE
#undef SCINT_ROWS_SHORT
#undef SCINT_ROWS
#undef SCINT_ROWS_K
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// cols: one DIF radix pass along the strided axis
// ------------------------------------------------------------------------------
struct ColPass {
    int ncols;          // contiguous extent handled by lanes
    int len;            // full FFT length R along the strided axis
    int block_len;      // Lp: current DIF block length
    int sub;            // S = Lp / radix
    const cplx* tw;     // W_len table
    int tw_mult;        // len / Lp
    // digit bookkeeping for the last pass: radices of the earlier passes (first pass
    // = least-significant digit of the output frequency index)
    int npre;
    int pre_radix[6];
    int last;           // 1 if this pass un-scrambles
};

// Loader: cplx ld(batch, row, col)   (row along the strided axis)
// Storer: void st(batch, row, col, v)  -- row is the in-place row for inner passes
//         and the NATURAL frequency index for the last pass.
// Rows, columns and strides are 32-bit; the functors widen once for the element offset.
template <int R, class Loader, class Storer>
__global__ void __launch_bounds__(256)
fft_cols_kernel(ColPass p, Loader ld, Storer st) {
    const int col = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (col >= p.ncols) return;
    const int bf = (int)blockIdx.y;        // butterfly id: b*S + s  (wave-uniform)
    const int64_t batch = blockIdx.z;
    const int b = bf / p.sub, s = bf - b * p.sub;
    const int row0 = b * p.block_len + s;
    cplx x[R];
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = ld(batch, row0 + m * p.sub, col);
    SmallFFT<R>::run(x);
    if (!p.last) {
        // twiddle W_Lp^{s k} (wave-uniform -> scalar loads), store in place
        const int step = s * p.tw_mult;
        st(batch, row0, col, x[0]);
#pragma unroll
        for (int k = 1; k < R; ++k) {
            const cplx w = p.tw[step * k];  // s*k < Lp  =>  index < len
            st(batch, row0 + k * p.sub, col, x[k] * w);
        }
    } else {
        // position digits of b (most significant first) are the earlier passes'
        // output indices k_1, k_2, ...; natural index = k_1 + R_1 k_2 + ... + (len/R) k_last
        int rem = b, scale = p.len / R, kbase = 0, weight = 1;
        // decode from the least-significant position digit (the latest pass) upward
        int digits[6];
        for (int i = p.npre - 1; i >= 0; --i) {
            digits[i] = rem % p.pre_radix[i];
            rem /= p.pre_radix[i];
        }
        for (int i = 0; i < p.npre; ++i) {
            kbase += digits[i] * weight;
            weight *= p.pre_radix[i];
        }
#pragma unroll
        for (int k = 0; k < R; ++k) st(batch, kbase + k * scale, col, x[k]);
    }
}

// Plan of column passes for a power-of-two length.
struct ColPlan {
    int npass;
    int radix[6];
};
inline ColPlan make_col_plan(int64_t len) {
    ColPlan pl;
    pl.npass = 0;
    int l = ilog2(len);
    // radix-16 passes first, the remainder (2/4/8) last; a lone remainder of 2 or 4
    // is merged into the previous pass as radix 32 when possible.
    while (l >= 4) { pl.radix[pl.npass++] = 16; l -= 4; }
    if (l > 0) {
        if (l == 1 && pl.npass > 0) pl.radix[pl.npass - 1] = 32;
        else pl.radix[pl.npass++] = 1 << l;
    }
    if (pl.npass == 0) { pl.radix[0] = 1; pl.npass = 1; }
    return pl;
}

template <int R, class Loader, class Storer>
int32_t launch_cols_pass(const ColPass& p, int64_t batches, Loader ld, Storer st,
                         hipStream_t stream) {
    const int block = p.ncols >= 256 ? 256 : (p.ncols >= 128 ? 128 : 64);
    dim3 grid((unsigned)ceil_div(p.ncols, block), (unsigned)(p.len / R), (unsigned)batches);
    SCINT_REQUIRE(p.len / R <= 65535 && batches <= 65535, "fft cols: grid too large");
    hipLaunchKernelGGL((fft_cols_kernel<R, Loader, Storer>), grid, dim3(block), 0, stream, p, ld, st);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ------------------------------------------------------------------------------
// cols, two-pass form: the strided-axis FFT of length L = L1 * L2 (16 <= L1, L2 <= 128) as the
// four-step algorithm on tiles of 16 adjacent columns, each step one trip through HBM:
//
//   pass A  for every n2 < L2:   Y[k1][n2] = W_L^{n2 k1} * sum_n1 x[n1 L2 + n2] W_L1^{n1 k1}
//           (rows n1 L2 + n2 of the source, L1-point FFTs, twiddle, stored at row k1 L2 + n2)
//   pass B  for every k1 < L1:   X[k1 + L1 k2] = sum_n2 Y[k1][n2] W_L2^{n2 k2}
//           (L2 CONSECUTIVE rows of the intermediate, L2-point FFTs, natural-order sink)
//
// Both passes are the row kernel above (one "slot" = one column of one sub-transform, living in
// LDS) behind transposing loaders: consecutive slots are consecutive columns and the threads of a
// slot are consecutive sub-transform indices, so one wave instruction touches 16 adjacent
// columns (256 contiguous bytes) of 4-8 rows -- whole cache lines in both directions.  The radix
// passes of fft_cols_kernel need log16(L) trips instead of two; run_cols_fft picks per size.
// Index arithmetic is 32-bit (rows, columns, strides and tile counts are far below 2^31; only the
// final element offset is 64-bit): a 64-bit multiply is four quarter-rate instructions on CDNA, and
// these kernels issue one address per 16-byte element.
struct TileSlot {
    int inner, col; int64_t batch; bool ok;
};
// Slot numbering of the tiled passes (c = column within the 16-column tile, fastest):
//   tile-fast (default)  slot = (((batch << lbits | inner) * ntiles + tile) * 16 + c
//       consecutive workgroups walk ALONG the rows of the array: the 64 slots of a workgroup are
//       64 adjacent columns (1 KiB per row) and the workgroups in flight together cover a few
//       contiguous multi-MiB row bands (DRAM pages and TLB entries get reused);
//   inner-fast           slot = (((batch * ntiles + tile) << lbits | inner) * 16 + c
//       a workgroup is one 256-byte tile of 4x as many rows (measured 1-2 % slower; the decode stays for the record).
__device__ inline TileSlot tile_slot(int64_t slot, int lbits, int ntiles, int ncols, int tile_fast) {
    TileSlot s;
    const uint32_t rest = (uint32_t)(slot >> 4), mask = (1u << lbits) - 1u;
    uint32_t tile, batch;
    if (tile_fast) {
        const uint32_t q = rest / (uint32_t)ntiles;
        tile = rest - q * (uint32_t)ntiles;
        s.inner = (int)(q & mask);
        batch = q >> lbits;
    } else {
        s.inner = (int)(rest & mask);
        const uint32_t tb = rest >> lbits;
        batch = tb / (uint32_t)ntiles;
        tile = tb - batch * (uint32_t)ntiles;
    }
    s.batch = batch;
    s.col = (int)tile * 16 + (int)(slot & 15);
    s.ok = s.col < ncols;
    return s;
}
template <class First>
struct ColsALoad {
    First first; int ncols, ntiles, tile_fast; int l2;        // L2 = 1 << l2
    struct Slot {
        const ColsALoad& p; TileSlot s;
        __device__ inline cplx operator()(int n1) const {
            if (!s.ok) return mk(0.0, 0.0);
            return p.first(s.batch, (n1 << p.l2) + s.inner, s.col);
        }
    };
    __device__ inline Slot open(int64_t slot) const { return Slot{*this, tile_slot(slot, l2, ntiles, ncols, tile_fast)}; }
};
template <class MidStorer>
struct ColsAStore {
    static constexpr bool kPair = false;
    MidStorer mid; int ncols, ntiles, tile_fast; int l2; const cplx* tw;   // tw = W_L table
    struct Slot {
        const ColsAStore& p; TileSlot s;
        __device__ inline void operator()(int k1, cplx v) const {
            if (!s.ok) return;
            p.mid(s.batch, (k1 << p.l2) + s.inner, s.col, v * p.tw[s.inner * k1]);
        }
    };
    __device__ inline Slot open(int64_t slot) const { return Slot{*this, tile_slot(slot, l2, ntiles, ncols, tile_fast)}; }
};
template <class MidLoader>
struct ColsBLoad {
    MidLoader mid; int ncols, ntiles, tile_fast; int l1, l2;
    struct Slot {
        const ColsBLoad& p; TileSlot s;
        __device__ inline cplx operator()(int n2) const {
            if (!s.ok) return mk(0.0, 0.0);
            return p.mid(s.batch, (s.inner << p.l2) + n2, s.col);
        }
    };
    __device__ inline Slot open(int64_t slot) const { return Slot{*this, tile_slot(slot, l1, ntiles, ncols, tile_fast)}; }
};
template <class LastStorer>
struct ColsBStore {
    static constexpr bool kPair = false;
    LastStorer last; int ncols, ntiles, tile_fast; int l1;
    struct Slot {
        const ColsBStore& p; TileSlot s;
        __device__ inline void operator()(int k2, cplx v) const {
            if (!s.ok) return;
            p.last(s.batch, s.inner + (k2 << p.l1), s.col, v);
        }
    };
    __device__ inline Slot open(int64_t slot) const { return Slot{*this, tile_slot(slot, l1, ntiles, ncols, tile_fast)}; }
};

// Runs a strided-axis FFT.  `first` loads the source, `mid_ld/mid_st` are the plain in-place
// accessors of the working array, `last` stores the final (natural-order) result.
// Lengths >= 256: two tiled passes (above); shorter ones: radix passes of fft_cols_kernel.
template <class FirstLoader, class MidLoader, class MidStorer, class LastStorer>
int32_t run_cols_fft(int64_t len, int64_t ncols, int64_t batches, FirstLoader first,
                     MidLoader mid_ld, MidStorer mid_st, LastStorer last, hipStream_t stream) {
    SCINT_REQUIRE(is_pow2(len) && len >= 2, "fft cols: length must be a power of two >= 2");
    SCINT_REQUIRE(len <= (1 << 24) && ncols < (1 << 30), "fft cols: extent beyond the 32-bit index range");
    const cplx* tw = twiddle_table(len);
    if (!tw) return SCINT_E_HIP;
    // Measured on MI355X: while the working array fits the 256 MiB Infinity Cache the in-place radix
    // passes re-read what they just wrote from the cache and win (4096^2 complex: 0.23 vs 0.28 ms);
    // beyond it every radix pass is an HBM round trip and the two tiled passes win (16384^2: 5.6
    // vs 7.3 ms).
    const bool beyond_cache = (double)len * (double)ncols * (double)batches * 16.0 > 300.0 * 1048576.0;
    if (beyond_cache && len >= 256 && len <= 16384) {
        const int l = ilog2(len), l2 = l / 2, l1 = l - l2;          // L1 >= L2, both in [16, 128]
        const int64_t ntiles = ceil_div(ncols, 16);
        constexpr int tile_fast = 1;     // workgroups walk along the rows of the array (measured: 1-2 % over tile columns)
        SCINT_REQUIRE(batches * ntiles * 128 < ((int64_t)1 << 32), "fft cols: too many tile slots");
        int32_t rc = launch_fft_rows<4, 7>((int64_t)1 << l1, batches * ntiles * ((int64_t)1 << l2) * 16,
                                     ColsALoad<FirstLoader>{first, (int)ncols, (int)ntiles, tile_fast, l2},
                                     ColsAStore<MidStorer>{mid_st, (int)ncols, (int)ntiles, tile_fast, l2, tw}, stream);
        if (rc != SCINT_OK) return rc;
        return launch_fft_rows<4, 7>((int64_t)1 << l2, batches * ntiles * ((int64_t)1 << l1) * 16,
                               ColsBLoad<MidLoader>{mid_ld, (int)ncols, (int)ntiles, tile_fast, l1, l2},
                               ColsBStore<LastStorer>{last, (int)ncols, (int)ntiles, tile_fast, l1}, stream);
    }
    ColPlan pl = make_col_plan(len);
    int Lp = (int)len;
    ColPass p;
    p.ncols = (int)ncols;
    p.len = (int)len;
    p.tw = tw;
    p.npre = 0;
    for (int i = 0; i < pl.npass; ++i) {
        const int R = pl.radix[i];
        p.block_len = Lp;
        p.sub = Lp / R;
        p.tw_mult = (int)len / Lp;
        p.last = (i == pl.npass - 1);
        int32_t rc = SCINT_OK;
#define SCINT_COLS(RR)                                                                    \
    if (i == 0 && p.last) rc = launch_cols_pass<RR>(p, batches, first, last, stream);      \
    else if (i == 0) rc = launch_cols_pass<RR>(p, batches, first, mid_st, stream);         \
    else if (p.last) rc = launch_cols_pass<RR>(p, batches, mid_ld, last, stream);          \
    else rc = launch_cols_pass<RR>(p, batches, mid_ld, mid_st, stream);
        switch (R) {
            case 2: SCINT_COLS(2) break;
            case 4: SCINT_COLS(4) break;
            case 8: SCINT_COLS(8) break;
            case 16: SCINT_COLS(16) break;
            case 32: SCINT_COLS(32) break;
            default: SCINT_REQUIRE(false, "fft cols: bad radix");
        }
#undef SCINT_COLS
        if (rc != SCINT_OK) return rc;
        p.pre_radix[p.npre++] = R;
        Lp /= R;
    }
    return SCINT_OK;
}

// ---- plain array accessors ------------------------------------------------------
struct ArrayLoad {   // a[batch][row][col]
    const cplx* a; int ld; int64_t batch_stride;
    __device__ inline cplx operator()(int64_t bt, int r, int c) const {
        return a[bt * batch_stride + ((int64_t)r * ld + c)];
    }
};
struct ArrayStore {
    cplx* a; int ld; int64_t batch_stride;
    __device__ inline void operator()(int64_t bt, int r, int c, cplx v) const {
        a[bt * batch_stride + ((int64_t)r * ld + c)] = v;
    }
};

}  // namespace scint
