// blockw.hpp -- W-vector block Lanczos: the step algebra (host + device, W a template argument) and
// the kernels of the W > 2 recurrence.
//
// STATUS: opt-in (SCINT_LANCZOS_BLOCK=4 / 8).  The algebra below is checked on the CPU against
// tools/models/blockw_reference.py (tests/test_blockw_cpu.py, tests/tools/blockw_host_check); the
// kernels that use it (blockw_kernels.hpp, blockq_kernels.hpp) are green on the host interpreter
// (tests/emu) and have NOT run on a GPU yet.  Nothing in the default path uses this file.
//
// Conventions (every matrix W x W complex unless noted), as in tools/models/blockw_reference.py:
//   W_j   = A Q_j - Q_{j-1} B_{j-1}^H           (N x W; mat-vec + reduce kernel)
//   A_j   = Q_j^H W_j                           (Hermitian; fixed-order partial sums)
//   G_j   = W_j^H W_j                           (Gram matrix; same partial sums)
//   G'_j  = G_j - A_j^H A_j = B_j^H B_j         (B_j upper triangular: Cholesky factor)
//   Q_{j+1} = (W_j - Q_j A_j) B_j^{-1}          (row by row, forward substitution over the columns)
//   T     = block tridiagonal, T[j][j] = A_j, T[j+1][j] = B_j -> Hermitian band matrix, half width W
//
// Packed coefficient ("W*W doubles"): the W real diagonal entries, then the strict upper triangle row
// by row as (re, im) pairs.  For W = 2 this is [d0, d1, re01, im01], the layout of the two-vector code.
#pragma once
#include <math.h>

#include "common.hpp"

namespace scint {

// relative size (against the column's Gram entry) below which a Cholesky pivot counts as zero; the
// rounding noise of G - A^H A - sum |B|^2 is a few 1e-16 of the Gram entry
constexpr double kBwPivotFloor = 1e-13;

template <int W>
__host__ __device__ constexpr int bw_upper(int r, int c) {   // index of Re of entry (r, c), r < c
    return W + 2 * (r * W - r * (r + 1) / 2 + (c - r - 1));
}

// Coefficients of one block step: A_{j-1} (Hermitian), B_{j-1} (upper triangular), 1 / diag(B_{j-1}).
// Only entries with r <= c are stored / meaningful.
template <int W>
struct BlkW {
    cplx a[W][W];
    cplx b[W][W];
    double inv[W];      // 0 when that direction is exhausted (non-positive pivot)
};

// A[r][c] for any (r, c) of the Hermitian A
template <int W>
__host__ __device__ inline cplx bw_a(const BlkW<W>& k, int r, int c) {
    return r <= c ? k.a[r][c] : conj(k.a[c][r]);
}

// From the SUMMED partials sa = pack(Q^H W), sg = pack(W^H W): A, the Cholesky factor B of
// G - A^H A, and the inverse pivots.
// `room`: directions the Krylov space can still take (n minus the rank of the blocks so far); what
// survives the pivot floor beyond that is the noise of a saturated space.
template <int W>
__host__ __device__ inline BlkW<W> bw_from_sums(const double* sa, const double* sg, int room = 1 << 30) {
    BlkW<W> k;
    cplx g[W][W];
#pragma unroll
    for (int r = 0; r < W; ++r) {
        k.a[r][r] = mk(sa[r], 0.0);
        g[r][r] = mk(sg[r], 0.0);
#pragma unroll
        for (int c = r + 1; c < W; ++c) {
            k.a[r][c] = mk(sa[bw_upper<W>(r, c)], sa[bw_upper<W>(r, c) + 1]);
            g[r][c] = mk(sg[bw_upper<W>(r, c)], sg[bw_upper<W>(r, c) + 1]);
        }
    }
    // H = G - A^H A (upper triangle incl. the diagonal); (A^H A)[r][c] = sum_m conj(A[m][r]) A[m][c]
    cplx h[W][W];
#pragma unroll
    for (int r = 0; r < W; ++r)
#pragma unroll
        for (int c = r; c < W; ++c) {
            cplx s = mk(0.0, 0.0);
#pragma unroll
            for (int m = 0; m < W; ++m) s = s + mulc(bw_a<W>(k, m, c), bw_a<W>(k, m, r));   // A[m][c] conj(A[m][r])
            h[r][c] = g[r][c] - s;
        }
    // Cholesky H = B^H B, B upper triangular, one row of B at a time
#pragma unroll
    for (int c = 0; c < W; ++c) {
        double d = h[c][c].x;
#pragma unroll
        for (int m = 0; m < c; ++m) d -= norm2(k.b[m][c]);
        // a pivot at the rounding level of the Gram entry it was subtracted from is a direction that
        // is exhausted (duplicate start rows, block wider than the matrix, converged directions): its
        // normalised column would be amplified noise, not orthogonal to the others
        const double piv = (room > 0 && d > kBwPivotFloor * g[c][c].x) ? sqrt(d) : 0.0;
        room -= piv > 0.0;
        k.b[c][c] = mk(piv, 0.0);
        k.inv[c] = piv > 0.0 ? 1.0 / piv : 0.0;
#pragma unroll
        for (int j = c + 1; j < W; ++j) {
            cplx s = h[c][j];
#pragma unroll
            for (int m = 0; m < c; ++m) s = s - mulc(k.b[m][j], k.b[m][c]);                    // conj(B[m][c]) B[m][j]
            k.b[c][j] = s * k.inv[c];
        }
    }
    return k;
}

// Row of Q_j = (W_{j-1} - Q_{j-1} A_{j-1}) B_{j-1}^{-1} from the rows u of W_{j-1} and q of Q_{j-1}.
template <int W>
__host__ __device__ inline void bw_q_row(const BlkW<W>& k, const cplx (&u)[W], const cplx (&q)[W], cplx (&x)[W]) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
        cplx y = u[c];
#pragma unroll
        for (int m = 0; m < W; ++m) y = y - q[m] * bw_a<W>(k, m, c);
#pragma unroll
        for (int m = 0; m < c; ++m) y = y - x[m] * k.b[m][c];
        x[c] = y * k.inv[c];
    }
}

// Row of Q_{j-1} B_{j-1}^H (what the reduce kernel subtracts from A Q_j): sum_{m >= c} q_m conj(B[c][m])
template <int W>
__host__ __device__ inline void bw_qbh_row(const BlkW<W>& k, const cplx (&q)[W], cplx (&out)[W]) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
        cplx s = mk(0.0, 0.0);
#pragma unroll
        for (int m = c; m < W; ++m) s = s + mulc(q[m], k.b[c][m]);
        out[c] = s;
    }
}

// ---- the band matrix T ------------------------------------------------------------
// band[i * (W + 1) + k] = T[i + k][i], k = 0..W (k = 0: the real diagonal in .x); n rows.
// Entry k of row r of block j from the packed A_j and the packed B_j (B_j couples blocks j and j + 1;
// nullptr for the last block).
template <int W>
__host__ __device__ inline cplx bw_band_entry(const double* A, const double* B, int r, int k) {
    if (k == 0) return mk(A[r], 0.0);
    if (r + k < W) {                                   // inside the diagonal block: A_j[r + k][r] = conj(A_j[r][r + k])
        const int idx = bw_upper<W>(r, r + k);
        return mk(A[idx], -A[idx + 1]);
    }
    if (!B) return mk(0.0, 0.0);
    const int rp = r + k - W;                          // row of B_j, rp <= r: entry B_j[rp][r]
    if (rp == r) return mk(B[r], 0.0);
    const int idx = bw_upper<W>(rp, r);
    return mk(B[idx], B[idx + 1]);
}

// packed forms of the coefficients of a step
template <int W>
__host__ __device__ inline void bw_pack(const BlkW<W>& k, double* pa, double* pb) {
#pragma unroll
    for (int r = 0; r < W; ++r) {
        pa[r] = k.a[r][r].x;
        pb[r] = k.b[r][r].x;
#pragma unroll
        for (int c = r + 1; c < W; ++c) {
            pa[bw_upper<W>(r, c)] = k.a[r][c].x; pa[bw_upper<W>(r, c) + 1] = k.a[r][c].y;
            pb[bw_upper<W>(r, c)] = k.b[r][c].x; pb[bw_upper<W>(r, c) + 1] = k.b[r][c].y;
        }
    }
}

// Block steps after which the Krylov space of a job is complete.  Block i contributes rank(Q_i) =
// the nonzero pivots of B_{i-1} (beta[i], packed) independent directions; once their sum reaches n
// the projection T is the whole matrix and later steps only add noise.  Returns min(k, that step
// count) and the accumulated rank.  A start block can be rank deficient (block wider than the
// matrix, dependent rows), so W k >= n alone proves nothing.
template <int W>
__host__ __device__ inline int bw_complete_steps(const double* beta, int k, int n, int* rank_out) {
    int rank = 0;
    for (int j = 0; j < k; ++j) {
        for (int r = 0; r < W; ++r) rank += beta[W * W * j + r] > 0.0;
        if (rank >= n) { *rank_out = rank; return j + 1; }
    }
    *rank_out = rank;
    return k;
}

// LDL^H pivots of T - x for a Hermitian band matrix of half width W: returns the number of negative
// pivots (eigenvalues below x).  Optionally stores the factor (d[i], m[i*W + k-1] = M_{i+k,i}, M = L D)
// for the inverse iteration.
template <int W>
__host__ __device__ inline int bw_band_count(const cplx* band, int n, double x, double tiny, double* d_out = nullptr,
                                             cplx* m_out = nullptr) {
    // sliding window over the last W columns: rinv[m-1] = 1 / d_{i-m}, win[m-1][k-1] = M_{i-m+k, i-m}
    double rinv[W];
    cplx win[W][W];
#pragma unroll
    for (int m = 0; m < W; ++m) {
        rinv[m] = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) win[m][k] = mk(0.0, 0.0);
    }
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const cplx* bi = band + (size_t)i * (W + 1);
        // d_i = T_ii - x - sum_m |M_{i,i-m}|^2 / d_{i-m};  M_{i,i-m} = win[m-1][m-1]
        double d = bi[0].x - x;
#pragma unroll
        for (int m = 1; m <= W; ++m) d -= norm2(win[m - 1][m - 1]) * rinv[m - 1];
        if (fabs(d) < tiny) d = -tiny;
        cnt += d < 0.0;
        // M_{i+k,i} = T_{i+k,i} - sum_{m>=1, m+k<=W} M_{i+k,i-m} conj(M_{i,i-m}) / d_{i-m}
        cplx col[W];
#pragma unroll
        for (int k = 1; k <= W; ++k) {
            cplx v = (i + k < n) ? bi[k] : mk(0.0, 0.0);
#pragma unroll
            for (int m = 1; m + k <= W; ++m) v = v - mulc(win[m - 1][m + k - 1], win[m - 1][m - 1]) * rinv[m - 1];
            col[k - 1] = v;
        }
        if (d_out) {
            d_out[i] = d;
#pragma unroll
            for (int k = 0; k < W; ++k) m_out[(size_t)i * W + k] = col[k];
        }
        // shift the window: column i becomes "i - 1"
#pragma unroll
        for (int m = W - 1; m >= 1; --m) {
            rinv[m] = rinv[m - 1];
#pragma unroll
            for (int k = 0; k < W; ++k) win[m][k] = win[m - 1][k];
        }
        rinv[0] = 1.0 / d;
#pragma unroll
        for (int k = 0; k < W; ++k) win[0][k] = col[k];
    }
    return cnt;
}

// Two steps of inverse iteration with the stored factor of T - sigma (d, m from bw_band_count):
// s <- (T - sigma)^{-1} s, scaled to max |component| = 1; returns sum |s_i|^2.  s must hold n entries;
// d is overwritten with its reciprocals.
template <int W>
__host__ __device__ inline double bw_inverse_iteration(double* d, const cplx* m, int n, cplx* s, int iters = 2) {
    // the pivots are only ever divided by: invert them once, in place (a division per band entry and
    // sweep would sit in the dependent chain of the substitutions, which one lane runs alone)
    for (int i = 0; i < n; ++i) d[i] = 1.0 / d[i];
    for (int i = 0; i < n; ++i) s[i] = mk(1.0, 0.0);
    double nrm = (double)n;
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < n; ++i) {                      // L y = s, L_{i,i-k} = M_{i,i-k} / d_{i-k}
            cplx y = s[i];
            for (int k = 1; k <= W && k <= i; ++k) y = y - (m[(size_t)(i - k) * W + k - 1] * s[i - k]) * d[i - k];
            s[i] = y;
        }
        for (int i = 0; i < n; ++i) s[i] = s[i] * d[i];
        for (int i = n - 1; i >= 0; --i) {                 // L^H z = y
            cplx z = s[i];
            for (int k = 1; k <= W && i + k < n; ++k) z = z - mulc(s[i + k], m[(size_t)i * W + k - 1]) * d[i];
            s[i] = z;
        }
        double mx = 0.0;
        for (int i = 0; i < n; ++i) mx = fmax(mx, fmax(fabs(s[i].x), fabs(s[i].y)));
        const double sc = (mx > 0.0 && isfinite(mx)) ? 1.0 / mx : 0.0;
        nrm = 0.0;
        for (int i = 0; i < n; ++i) { s[i] = s[i] * sc; nrm += norm2(s[i]); }
    }
    return nrm;
}

// || B s_last ||^2 for the last block of the eigenvector (B upper triangular, from a BlkW)
template <int W>
__host__ __device__ inline double bw_resid2(const BlkW<W>& k, const cplx* s_last) {
    double r2 = 0.0;
#pragma unroll
    for (int r = 0; r < W; ++r) {
        cplx v = mk(0.0, 0.0);
#pragma unroll
        for (int c = r; c < W; ++c) v = v + k.b[r][c] * s_last[c];
        r2 += norm2(v);
    }
    return r2;
}

}  // namespace scint
