// blockq_kernels.hpp -- wide-block Lanczos (W = 8; W = 4 for cross-checks): the "q" kernel family.
//
// STATUS: opt-in (SCINT_LANCZOS_BLOCK=8, or =4 with SCINT_MATVEC_MFMA=2), validated on the host
// interpreter only (tests/emu); NEVER RUN ON A GPU.  Nothing in the default path uses this file.
//
// Why a second family.  The pk2_* / pkw_* kernels never store Q_j ahead of its use: every consumer
// (each mat-vec workgroup, each reduce block, the check) re-derives the coefficients of the step
// from the fixed-order partial sums -- 2 W^2 wave reductions -- and rebuilds the rows of
// Q_j = (W_{j-1} - Q_{j-1} A_{j-1}) B_{j-1}^{-1} it needs.  That is 8 reductions and a 2x2 solve
// for two vectors, but 128 reductions and 264 coefficient registers for eight.  Here a step is four
// launches instead of two, and nothing is computed twice:
//   pkq_coef_kernel    one wave per curvature: A_{j-1} = sum of the partials of Q^H W, the Cholesky
//                      factor B_{j-1} of W^H W - A^H A, 1 / diag(B); also the history the check reads
//   pkq_qbuild_kernel  Q_j, row by row, written once (3 N W complex of traffic: ~1 % of a pass)
//   pkq_matvec_band_kernel / pkq_matvec_mfma_kernel   the matrix-core mat-vec of blockw_kernels.hpp
//                      with X_J / X_I simply copied from Q_j (no coefficients, no rebuild), over bands of
//                      R block rows (default for eight vectors) or plain strips
//   pkq_reduce_kernel  W_j = A Q_j - Q_{j-1} B_{j-1}^H and the partials of the next coefficients
// The convergence check is the same algorithm as pkw_check_kernel (64-shift multisection on the
// banded LDL^H Sturm count, inverse iteration for the residual) with the per-lane W x W window of
// the factorisation in LDS instead of registers (64 complex per lane do not fit).
// With the same 16 x 16 x 4 matrix-core instruction stream, W = 8 fills all 16 B-columns (W = 4
// leaves half of them zero): CPU model of the passes at N = 4095 -- 31.9 (W = 2), 24.6 (4), 19.9 (8).
#pragma once
#include "blockw_kernels.hpp"

namespace scint {

// block steps the check kernel holds in LDS: 64 (T up to 512 x 512 for eight vectors: 155 KiB of the
// 160 KiB of LDS); 128 for two vectors, as the pk2 kernels allow
template <int W> constexpr int kq_max_steps() { return W <= 2 ? 128 : 64; }
constexpr int kRedGroupsQ = 4;      // wavefronts per reduce block (LDS: groups x 64 x W complex)
// tiles per strip of the mat-vec (its X_J blocks live in LDS): 16 / 8 for two / four vectors; for eight vectors 4
// (72 KiB, two workgroups per CU) or, with SCINT_Q_STRIP=8, 8 (104 KiB, one workgroup per CU, half the
// X_J / row-partial traffic) -- to be decided by measurement
template <int W> struct QShape { static constexpr int strip = W >= 8 ? 4 : (W >= 4 ? 8 : 16); };

// layout of PackedJob::coef (doubles): A full [W][W] complex | B upper [W][W] complex (zeros below)
// | 1/diag(B) [W] | packed A [W*W] | packed B [W*W]
template <int W> struct QCoef {
    static constexpr int S = W * W;
    static constexpr int a_full = 0, b_up = 2 * S, inv = 4 * S, a_pack = 4 * S + W, b_pack = 5 * S + W, total = 6 * S + W;
};

template <int W>
__global__ void __launch_bounds__(64) pkq_coef_kernel(const PackedJob* jobs, int launch) {
    constexpr int S = W * W;
    typedef QCoef<W> C;
    __shared__ double sa[S], sg[S], pb[S];
    __shared__ cplx A[W][W], H[W][W], B[W][W];
    __shared__ double inv[W];
    const PackedJob jb = jobs[blockIdx.x];
    const int step = launch - jb.start;
    // step == max_steps is legal: the coefficients of the LAST completed step, for the check
    if (jb.gen <= 0 || jb.n < 2 || step < 0 || step > jb.max_steps || gload(jb.state) >= jb.gen) return;
    const int par = step & 1, lane = threadIdx.x;
    const double* __restrict__ ap = par ? jb.apart[1] : jb.apart[0];
    const double* __restrict__ up = par ? jb.upart[1] : jb.upart[0];
    // fixed order: lane c owns scalar c and walks the 64-row blocks in order
    // (eight blocks' loads are issued before their sums are taken, in block order: a loop of load-wait-add
    // would pay the memory latency nb times)
    for (int c = lane; c < S; c += 64) {
        double a = 0.0, g = 0.0;
        for (int i0 = 0; i0 < jb.nb; i0 += 8) {
            double xa[8], xg[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, jb.nb - 1);
                xa[u] = gload(ap + S * i + c); xg[u] = gload(up + S * i + c);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < jb.nb) { a += xa[u]; g += xg[u]; }
        }
        sa[c] = a; sg[c] = g;
    }
    // directions the Krylov space has left: n minus the nonzero pivots of the blocks so far (lanes in parallel)
    __shared__ int room_s;
    {
        double used = 0.0;
        for (int idx = lane; idx < step * W; idx += 64) used += gload(jb.beta + S * (idx / W) + idx % W) > 0.0 ? 1.0 : 0.0;
        used = wave_sum(used);
        if (lane == 0) room_s = jb.n - (int)used;
    }
    __syncthreads();
    for (int idx = lane; idx < S; idx += 64) {
        const int r = idx / W, c = idx - r * W;
        if (r == c) { A[r][r] = mk(sa[r], 0.0); H[r][r] = mk(sg[r], 0.0); }
        else if (r < c) {
            const int u = bw_upper<W>(r, c);
            A[r][c] = mk(sa[u], sa[u + 1]); A[c][r] = mk(sa[u], -sa[u + 1]);
            H[r][c] = mk(sg[u], sg[u + 1]); H[c][r] = mk(sg[u], -sg[u + 1]);
        }
    }
    __syncthreads();
    // H = G - A^H A;  (A^H A)[r][c] = sum_m conj(A[m][r]) A[m][c]
    for (int idx = lane; idx < S; idx += 64) {
        const int r = idx / W, c = idx - r * W;
        cplx s = mk(0.0, 0.0);
        for (int m = 0; m < W; ++m) s = s + mulc(A[m][c], A[m][r]);
        H[r][c] = H[r][c] - s;
        B[r][c] = mk(0.0, 0.0);
    }
    __syncthreads();
    if (lane == 0) {   // Cholesky H = B^H B, B upper triangular, one row of B at a time (as bw_from_sums)
        // The Krylov space holds at most n directions: once the blocks so far (their nonzero pivots,
        // bw_complete_steps) and this one have n, whatever else survives the pivot floor is the
        // noise of a saturated space and would enter T as a non-orthogonal column.
        int room = room_s;
        for (int c = 0; c < W; ++c) {
            double d = H[c][c].x;
            for (int m = 0; m < c; ++m) d -= norm2(B[m][c]);
            const double piv = (room > 0 && d > kBwPivotFloor * sg[c]) ? sqrt(d) : 0.0;        // see bw_from_sums
            room -= piv > 0.0;
            B[c][c] = mk(piv, 0.0);
            inv[c] = piv > 0.0 ? 1.0 / piv : 0.0;
            for (int j = c + 1; j < W; ++j) {
                cplx s = H[c][j];
                for (int m = 0; m < c; ++m) s = s - mulc(B[m][j], B[m][c]);      // conj(B[m][c]) B[m][j]
                B[c][j] = s * inv[c];
            }
        }
        for (int r = 0; r < W; ++r) {
            pb[r] = B[r][r].x;
            for (int c = r + 1; c < W; ++c) { pb[bw_upper<W>(r, c)] = B[r][c].x; pb[bw_upper<W>(r, c) + 1] = B[r][c].y; }
        }
    }
    __syncthreads();
    double* __restrict__ co = jb.coef;
    for (int idx = lane; idx < S; idx += 64) {
        const int r = idx / W, c = idx - r * W;
        co[C::a_full + 2 * idx] = A[r][c].x; co[C::a_full + 2 * idx + 1] = A[r][c].y;
        co[C::b_up + 2 * idx] = B[r][c].x; co[C::b_up + 2 * idx + 1] = B[r][c].y;
        co[C::a_pack + idx] = sa[idx];                       // the summed partials ARE the packed A
        co[C::b_pack + idx] = pb[idx];
        // history for the check: A_{step-1} and B_{step-1} (B[step] couples blocks step-1 and step)
        if (step > 0) jb.alpha[S * (step - 1) + idx] = sa[idx];
        jb.beta[S * step + idx] = pb[idx];
    }
    if (lane < W) co[C::inv + lane] = inv[lane];
}

// Q_j = (W_{j-1} - Q_{j-1} A_{j-1}) B_{j-1}^{-1}, one 64-row block per wavefront
template <int W>
__global__ void __launch_bounds__(64) pkq_qbuild_kernel(const PackedJob* jobs, int launch) {
    constexpr int S = W * W;
    typedef QCoef<W> C;
    __shared__ double co[4 * S + W];
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x, e = threadIdx.x;
    const int step = launch - jb.start;
    if (K >= jb.nb || jb.n < 2 || step < 0 || step >= jb.max_steps || gload(jb.state) >= jb.gen) return;
    for (int i = e; i < 4 * S + W; i += 64) co[i] = gload(jb.coef + i);
    __syncthreads();
    const cplx* A = (const cplx*)(co + C::a_full);
    const cplx* B = (const cplx*)(co + C::b_up);
    const double* inv = co + C::inv;
    const int par = step & 1, qs = jb.qslots, r = K * kTB + e;
    const cplx* __restrict__ Up = par ? jb.U[1] : jb.U[0];
    const cplx* __restrict__ Qp = jb.Q + (int64_t)((step + qs - 1) % qs) * jb.qstride * W;
    cplx* __restrict__ Qn = jb.Q + (int64_t)(step % qs) * jb.qstride * W;
    cplx u[W], q[W], x[W];
#pragma unroll
    for (int v = 0; v < W; ++v) { u[v] = gload(Up + W * r + v); q[v] = gload(Qp + W * r + v); }
#pragma unroll
    for (int c = 0; c < W; ++c) {
        cplx y = u[c];
#pragma unroll
        for (int m = 0; m < W; ++m) y = y - q[m] * A[m * W + c];
#pragma unroll
        for (int m = 0; m < c; ++m) y = y - x[m] * B[m * W + c];
        x[c] = y * inv[c];
    }
#pragma unroll
    for (int v = 0; v < W; ++v) gstore(Qn + W * r + v, x[v]);
}

// cooperative copy of `count` complex values (count <= MAXN) from global memory into LDS by 256 threads:
// every thread issues all its loads before the first LDS store (a load-store loop would pay the memory
// latency once per 256 values); loads beyond `count` re-read the last element and are dropped
template <int MAXN>
__device__ inline void copy_to_lds_256(cplx* __restrict__ dst, const cplx* __restrict__ src, int count, int tid) {
    constexpr int U = (MAXN + 255) / 256;
    cplx tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tmp[u] = gload(src + min(tid + 256 * u, count - 1));
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (tid + 256 * u < count) dst[tid + 256 * u] = tmp[u];
}

// The matrix-core mat-vec (see pkw_matvec_mfma_kernel for the operand algebra) with X_J and X_I copied
// from the stored Q_j.  Dynamic LDS: xs[STRIP][64][2W] | xI[64][2W] | cred[4][64][2W] doubles.
template <int W, int STRIP> constexpr size_t pkq_matvec_lds_bytes() {
    return sizeof(double) * (size_t)(STRIP + 1 + 4) * kTB * 2 * W;
}
template <int W, int STRIP>
__global__ void __launch_bounds__(256, 2)
pkq_matvec_mfma_kernel(const PackedJob* __restrict__ jobs, const Strip* __restrict__ strips, int launch) {
    constexpr int NR = 2 * W;
    static_assert(NR <= 16 && (NR & (NR - 1)) == 0, "block width must be 1, 2, 4 or 8");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double (*xs)[kTB][NR] = (double (*)[kTB][NR])smem_raw;
    double (*xI)[NR] = (double (*)[NR])(smem_raw + sizeof(double) * STRIP * kTB * NR);
    double (*cred)[kTB][NR] = (double (*)[kTB][NR])(smem_raw + sizeof(double) * (STRIP + 1) * kTB * NR);
    const Strip st = strips[blockIdx.x];
    const PackedJob* __restrict__ jp = jobs + st.job;
    const int step = launch - jp->start;
    if (jp->n < 2 || step < 0 || step >= jp->max_steps || gload(jp->state) >= jp->gen) return;
    const int nb = jp->nb;
    const cplx* __restrict__ Qj = jp->Q + (int64_t)(step % jp->qslots) * jp->qstride * W;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k4 = lane >> 4, n16 = lane & 15;
    const bool odd = n16 & 1;
    const double live = n16 < NR ? 1.0 : 0.0;
    const int npair = (n16 & (NR - 1)) & ~1;
    const int I = st.I;
    const int64_t t0 = tile_offset(nb, I);
    const int ntile = st.J1 - st.J0;
    const cplx* __restrict__ tb = jp->tiles + (t0 + (st.J0 - I)) * kTileElems;
    const cplx* __restrict__ rp = tb + (16 * w + n16) * kTB + k4;
    const cplx* __restrict__ cp = tb + (16 * w + k4) * kTB + n16;
    cplx ra[16], ca[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) ra[g] = gload(rp + 4 * g);
    // rows of Q_j are [row][W] complex = [row][2 W] doubles: the blocks are plain copies
    copy_to_lds_256<STRIP * kTB * W>((cplx*)smem_raw, Qj + (int64_t)st.J0 * kTB * W, ntile * kTB * W, threadIdx.x);
    copy_to_lds_256<kTB * W>((cplx*)&xI[0][0], Qj + (int64_t)I * kTB * W, kTB * W, threadIdx.x);
    lds_barrier();
    double y1[4], y3[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double pr = xI[16 * w + 4 * kk + k4][npair], pi = xI[16 * w + 4 * kk + k4][npair + 1];
        y1[kk] = live * (odd ? pi : pr);
        y3[kk] = live * (odd ? -pr : pi);
    }
    v4d accr0 = {0.0, 0.0, 0.0, 0.0}, accr1 = {0.0, 0.0, 0.0, 0.0};
    double* __restrict__ colpart = (double*)jp->colpart;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const int64_t toff = (int64_t)t * kTileElems;
#pragma unroll
        for (int q = 0; q < 16; ++q) ca[q] = gload(cp + toff + (4 * (q & 3)) * kTB + 16 * (q >> 2));
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const double pr = xs[t][4 * g + k4][npair], pi = xs[t][4 * g + k4][npair + 1];
            const double x1 = live * (odd ? pi : pr);
            const double x2 = live * (odd ? pr : -pi);
            if (g & 1) { accr1 = mfma_f64_16x16x4(ra[g].x, x1, accr1); accr1 = mfma_f64_16x16x4(ra[g].y, x2, accr1); }
            else       { accr0 = mfma_f64_16x16x4(ra[g].x, x1, accr0); accr0 = mfma_f64_16x16x4(ra[g].y, x2, accr0); }
        }
        {   // row patches of the next tile (of this one again after the last: an unconditional load keeps
            // the compiler's wait counts exact)
            const int64_t noff = t + 1 < ntile ? toff + kTileElems : toff;
#pragma unroll
            for (int g = 0; g < 16; ++g) ra[g] = gload(rp + noff + 4 * g);
        }
        v4d accc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                accc[c] = mfma_f64_16x16x4(ca[4 * c + kk].x, y1[kk], accc[c]);
                accc[c] = mfma_f64_16x16x4(ca[4 * c + kk].y, y3[kk], accc[c]);
            }
        }
        if (n16 < NR) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) cred[w][16 * c + mfma_d_row(lane, r)][n16] = accc[c][r];
        }
        lds_barrier();
        const int Jt = st.J0 + t;
        if (Jt != I) {
            for (int idx = threadIdx.x; idx < kTB * NR; idx += 256) {
                const int col = idx / NR, nn = idx - col * NR;
                const double sum = ((cred[0][col][nn] + cred[1][col][nn]) + cred[2][col][nn]) + cred[3][col][nn];
                gstore(colpart + NR * ((t0 + (Jt - I)) * kTB + col) + nn, sum);
            }
        }
        lds_barrier();
    }
    double* __restrict__ rowpart = (double*)jp->rowpart;
    if (n16 < NR) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            gstore(rowpart + NR * ((int64_t)st.index * kTB + 16 * w + mfma_d_row(lane, r)) + n16, accr0[r] + accr1[r]);
    }
}

// the last stage of both reduce kernels (wave 0 of the block): W_j = total - Q_{j-1} B_{j-1}^H for the
// block's 64 rows and the partials of the next coefficients
template <int W>
__device__ __forceinline__ void pkq_reduce_tail(const PackedJob& jb, int K, int step, int par, int e,
                                       cplx (*part)[kTB][W], cplx (*Bs)[W]) {
    constexpr int S = W * W;
    const int qs = jb.qslots, r = K * kTB + e;
    const cplx* __restrict__ Qp = jb.Q + (int64_t)((step + qs - 1) % qs) * jb.qstride * W;
    const cplx* __restrict__ Qj = jb.Q + (int64_t)(step % qs) * jb.qstride * W;
    cplx* __restrict__ Un = par ? jb.U[0] : jb.U[1];
    cplx q[W], x[W], t[W];
#pragma unroll
    for (int v = 0; v < W; ++v) { q[v] = gload(Qp + W * r + v); x[v] = gload(Qj + W * r + v); }
#pragma unroll
    for (int c = 0; c < W; ++c) {
        cplx tot = part[0][e][c];
#pragma unroll
        for (int k = 1; k < kRedGroupsQ; ++k) tot = tot + part[k][e][c];
        cplx s = mk(0.0, 0.0);                        // row of Q_{j-1} B_{j-1}^H: sum_{m >= c} q_m conj(B[c][m])
#pragma unroll
        for (int m = c; m < W; ++m) s = s + mulc(q[m], Bs[c][m]);
        t[c] = tot - s;                               // row of W_j = A Q_j - Q_{j-1} B_{j-1}^H
        gstore(Un + W * r + c, t[c]);
    }
    // packed partials of A_j = Q_j^H W_j and of W_j^H W_j, written as they are produced
    double* __restrict__ an = (par ? jb.apart[0] : jb.apart[1]) + S * K;
    double* __restrict__ un = (par ? jb.upart[0] : jb.upart[1]) + S * K;
#pragma unroll
    for (int a = 0; a < W; ++a) {
        const double da = wave_sum(x[a].x * t[a].x + x[a].y * t[a].y);
        const double dg = wave_sum(norm2(t[a]));
        if (e == 0) { gstore(an + a, da); gstore(un + a, dg); }
#pragma unroll
        for (int b = a + 1; b < W; ++b) {
            const cplx za = wave_sum(mulc(t[b], x[a]));            // conj(x_a) t_b
            const cplx zg = wave_sum(mulc(t[b], t[a]));            // conj(t_a) t_b
            if (e == 0) {
                gstore(an + bw_upper<W>(a, b), za.x); gstore(an + bw_upper<W>(a, b) + 1, za.y);
                gstore(un + bw_upper<W>(a, b), zg.x); gstore(un + bw_upper<W>(a, b) + 1, zg.y);
            }
        }
    }
}

// ---- banded form: R block rows per workgroup -------------------------------------------------------
// The strip kernel above pays per TILE for what could be paid per group of tiles: one column partial
// (64 x W complex) written and read back, and the copy of X_J.  At W = 8 that is 8 KiB + 8 KiB + 8 KiB
// per 64-KiB tile; with the row partials and the X_I copies the extra traffic is ~0.47 of the matrix
// stream (model: 65 MB on 133 MB per pass at N = 4095), which would eat most of the 0.62x passes.  The
// matrix cores make the cure cheap: a row block's accumulator is 8 registers, so one workgroup can
// take R block rows (I0 .. I0+R-1) of a column chunk J0..J1: for every column J the column parts of
// the R tiles accumulate in the same four accumulators before ONE cross-wave reduction and ONE partial
// per (band, column); X_J is copied once for R tiles.  R = 4, chunks of 4: extra traffic 0.21.
// Tile columns left of a row's diagonal do not exist (packed upper triangle): row r takes part from
// column I0 + r on; the chunk grid starts at J0 = I0, and R <= STRIP makes every workgroup of a band
// produce a row partial for every row of the band.
// Dynamic LDS: xs[STRIP][64][2W] | xI[R][64][2W] | cred[4][64][min(2W, 8)] doubles.
template <int W, int STRIP, int R> constexpr size_t pkq_band_lds_bytes() {
    return sizeof(double) * ((size_t)(STRIP + R) * kTB * 2 * W + (size_t)4 * kTB * (2 * W > 8 ? 8 : 2 * W));
}
template <int W, int STRIP, int R>
__global__ void __launch_bounds__(256, 2)
pkq_matvec_band_kernel(const PackedJob* __restrict__ jobs, const Strip* __restrict__ strips, int launch) {
    constexpr int NR = 2 * W, NRH = NR > 8 ? 8 : NR, PHASES = NR / NRH;
    static_assert(NR <= 16 && (NR & (NR - 1)) == 0 && R <= STRIP, "bad band shape");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double (*xs)[kTB][NR] = (double (*)[kTB][NR])smem_raw;
    double (*xI)[kTB][NR] = (double (*)[kTB][NR])(smem_raw + sizeof(double) * STRIP * kTB * NR);
    double (*cred)[kTB][NRH] = (double (*)[kTB][NRH])(smem_raw + sizeof(double) * (STRIP + R) * kTB * NR);
    const Strip st = strips[blockIdx.x];
    const PackedJob* __restrict__ jp = jobs + st.job;
    const int step = launch - jp->start;
    if (jp->n < 2 || step < 0 || step >= jp->max_steps || gload(jp->state) >= jp->gen) return;
    const int nb = jp->nb;
    const cplx* __restrict__ Qj = jp->Q + (int64_t)(step % jp->qslots) * jp->qstride * W;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k4 = lane >> 4, n16 = lane & 15;
    const bool odd = n16 & 1;
    const double live = n16 < NR ? 1.0 : 0.0;
    const int npair = (n16 & (NR - 1)) & ~1;
    const int I0 = st.I, J0 = st.J0;
    const int nrow = min(R, nb - I0), ntile = st.J1 - J0;
    const cplx* __restrict__ tiles = jp->tiles;
    // tile (I0 + r, J0 + t) of the packed upper triangle (exists iff I0 + r <= J0 + t)
    auto tile_at = [&](int t, int r) { return tiles + (tile_offset(nb, I0 + r) + (J0 + t - I0 - r)) * kTileElems; };
    const int row_off = (16 * w + n16) * kTB + k4;       // row-part operand g:  + 4 g
    const int col_off = (16 * w + k4) * kTB + n16;       // column-part operand (c, kk): + 4 kk * 64 + 16 c
    cplx ra[16], ca[16];
    {
        const cplx* __restrict__ p0 = tile_at(0, 0) + row_off;
#pragma unroll
        for (int g = 0; g < 16; ++g) ra[g] = gload(p0 + 4 * g);
    }
    copy_to_lds_256<STRIP * kTB * W>((cplx*)smem_raw, Qj + (int64_t)J0 * kTB * W, ntile * kTB * W, threadIdx.x);
    copy_to_lds_256<R * kTB * W>((cplx*)&xI[0][0][0], Qj + (int64_t)I0 * kTB * W, nrow * kTB * W, threadIdx.x);
    lds_barrier();
    v4d accr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) accr[r] = (v4d){0.0, 0.0, 0.0, 0.0};
    double* __restrict__ colpart = (double*)jp->colpart;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const int J = J0 + t;
        v4d accc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r >= nrow || I0 + r > J) continue;                 // wave-uniform
            const cplx* __restrict__ tp = tile_at(t, r);
            const bool offdiag = I0 + r < J;
            // (loads are issued unconditionally -- the diagonal tile's column patches are read and not
            // used, the last tile's "next" is itself: a load under a branch makes the compiler's
            // wait counts pessimistic for everything queued behind it, and the pipeline collapses)
#pragma unroll
            for (int q = 0; q < 16; ++q) ca[q] = gload(tp + col_off + (4 * (q & 3)) * kTB + 16 * (q >> 2));   // q = 4 c + kk
            v4d a0 = accr[r], a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const double pr = xs[t][4 * g + k4][npair], pi = xs[t][4 * g + k4][npair + 1];
                const double x1 = live * (odd ? pi : pr);          // a x = (ar xr - ai xi) + i (ar xi + ai xr)
                const double x2 = live * (odd ? pr : -pi);
                if (g & 1) { a1 = mfma_f64_16x16x4(ra[g].x, x1, a1); a1 = mfma_f64_16x16x4(ra[g].y, x2, a1); }
                else       { a0 = mfma_f64_16x16x4(ra[g].x, x1, a0); a0 = mfma_f64_16x16x4(ra[g].y, x2, a0); }
            }
            accr[r] = a0 + a1;
            // row patches of the next tile of the sequence: the next row of this column, else row 0 of the next one
            {
                int tn = t, rn = r + 1;
                if (rn >= nrow || I0 + rn > J) { rn = 0; tn = t + 1; }
                const cplx* __restrict__ pn = (tn < ntile ? tile_at(tn, rn) : tp) + row_off;
#pragma unroll
                for (int g = 0; g < 16; ++g) ra[g] = gload(pn + 4 * g);
            }
            if (offdiag) {
                // B operands of the column part: rows 16 w + 4 kk + k4 of X_{I0 + r};  conj(a) x = (ar xr + ai xi) + i (ar xi - ai xr)
                double y1[4], y3[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double pr = xI[r][16 * w + 4 * kk + k4][npair], pi = xI[r][16 * w + 4 * kk + k4][npair + 1];
                    y1[kk] = live * (odd ? pi : pr);
                    y3[kk] = live * (odd ? -pr : pi);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        accc[c] = mfma_f64_16x16x4(ca[4 * c + kk].x, y1[kk], accc[c]);
                        accc[c] = mfma_f64_16x16x4(ca[4 * c + kk].y, y3[kk], accc[c]);
                    }
                }
            }
        }
        if (J > I0) {
            // cross-wave sum of the column's partial (all rows of the band), fixed order, one half of the
            // real columns per phase
#pragma unroll
            for (int ph = 0; ph < PHASES; ++ph) {
                if (n16 < NR && (n16 / NRH) == ph) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) cred[w][16 * c + mfma_d_row(lane, r4)][n16 - ph * NRH] = accc[c][r4];
                }
                lds_barrier();
                for (int idx = threadIdx.x; idx < kTB * NRH; idx += 256) {
                    const int col = idx / NRH, nn = idx - col * NRH;
                    const double sum = ((cred[0][col][nn] + cred[1][col][nn]) + cred[2][col][nn]) + cred[3][col][nn];
                    gstore(colpart + NR * ((tile_offset(nb, I0) + (J - I0)) * kTB + col) + ph * NRH + nn, sum);
                }
                lds_barrier();
            }
        }
    }
    double* __restrict__ rowpart = (double*)jp->rowpart;
    if (n16 < NR) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r >= nrow) continue;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                gstore(rowpart + NR * (((int64_t)st.index * R + r) * kTB + 16 * w + mfma_d_row(lane, r4)) + n16, accr[r][r4]);
        }
    }
}

// reduce for the banded mat-vec: block row K = R b + r sums the row partials (slot R * workgroup + r)
// of the workgroups of band b -- row_strip0[b] .. row_strip0[b + 1] -- and the column partials of
// the bands that start above it (I0' = R b' < K), stored at the tile slot (I0', K)
template <int W, int R>
__global__ void __launch_bounds__(64 * kRedGroupsQ)
pkq_reduce_band_kernel(const PackedJob* __restrict__ jobs, int launch) {
    constexpr int S = W * W;
    typedef QCoef<W> C;
    __shared__ cplx part[kRedGroupsQ][kTB][W];
    __shared__ cplx Bs[W][W];
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x;
    const int step = launch - jb.start;
    if (K >= jb.nb || jb.n < 2 || step < 0 || step >= jb.max_steps || gload(jb.state) >= jb.gen) return;
    const int par = step & 1;
    const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
    const int b = K / R, r = K - R * b;
    const int w0 = jb.row_strip0[b], nwg = jb.row_strip0[b + 1] - w0;
    const int nabove = (K + R - 1) / R;
    cplx acc[W];
#pragma unroll
    for (int v = 0; v < W; ++v) acc[v] = mk(0.0, 0.0);
    for (int idx = g; idx < nwg + nabove; idx += kRedGroupsQ) {
        const int I0p = R * (idx - nwg);
        const cplx* src = idx < nwg ? jb.rowpart + W * (((int64_t)(w0 + idx) * R + r) * kTB + e)
                                    : jb.colpart + W * ((tile_offset(jb.nb, I0p) + (K - I0p)) * kTB + e);
#pragma unroll
        for (int v = 0; v < W; ++v) acc[v] = acc[v] + gload(src + v);
    }
#pragma unroll
    for (int v = 0; v < W; ++v) part[g][e][v] = acc[v];
    for (int i = threadIdx.x; i < S; i += 64 * kRedGroupsQ)
        Bs[i / W][i % W] = mk(gload(jb.coef + C::b_up + 2 * i), gload(jb.coef + C::b_up + 2 * i + 1));
    __syncthreads();
    if (g == 0) pkq_reduce_tail<W>(jb, K, step, par, e, part, Bs);
}

template <int W>
__global__ void __launch_bounds__(64 * kRedGroupsQ)
pkq_reduce_kernel(const PackedJob* __restrict__ jobs, int launch) {
    constexpr int S = W * W;
    typedef QCoef<W> C;
    __shared__ cplx part[kRedGroupsQ][kTB][W];
    __shared__ cplx Bs[W][W];                       // B_{j-1}, upper triangular
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x;
    const int step = launch - jb.start;
    if (K >= jb.nb || jb.n < 2 || step < 0 || step >= jb.max_steps || gload(jb.state) >= jb.gen) return;
    const int par = step & 1;
    const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
    // fixed summation order: the strips of block row K, then the column partials of the tiles (cI, K), cI < K
    const int s0 = jb.row_strip0[K], nrow = jb.row_strip0[K + 1] - s0;
    cplx acc[W];
#pragma unroll
    for (int v = 0; v < W; ++v) acc[v] = mk(0.0, 0.0);
    for (int idx = g; idx < nrow + K; idx += kRedGroupsQ) {
        const int cI = idx - nrow;
        const cplx* src = idx < nrow ? jb.rowpart + W * ((int64_t)(s0 + idx) * kTB + e)
                                     : jb.colpart + W * ((tile_offset(jb.nb, cI) + (K - cI)) * kTB + e);
#pragma unroll
        for (int v = 0; v < W; ++v) acc[v] = acc[v] + gload(src + v);
    }
#pragma unroll
    for (int v = 0; v < W; ++v) part[g][e][v] = acc[v];
    for (int i = threadIdx.x; i < S; i += 64 * kRedGroupsQ)
        Bs[i / W][i % W] = mk(gload(jb.coef + C::b_up + 2 * i), gload(jb.coef + C::b_up + 2 * i + 1));
    __syncthreads();
    if (g == 0) pkq_reduce_tail<W>(jb, K, step, par, e, part, Bs);
}

// LDL^H pivots of T - x (bw_band_count) with the sliding window of the last W columns in memory:
// column i lives in slot i % W, element (slot, k) at win[(slot * W + k) * stride], 1 / d at
// rinv[slot * stride].  stride = 64 interleaves the windows of the 64 lanes (conflict-free LDS).
template <int W>
__device__ __forceinline__ int bq_band_count(const cplx* band, int n, double x, double tiny, cplx* win, double* rinv, int stride,
                                    double* d_out = nullptr, cplx* m_out = nullptr) {
    for (int s = 0; s < W; ++s) {
        rinv[s * stride] = 0.0;
        for (int k = 0; k < W; ++k) win[(s * W + k) * stride] = mk(0.0, 0.0);
    }
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const cplx* bi = band + (size_t)i * (W + 1);
        // column i - m sits in slot (i - m) mod W = (i + W - m) % W; M_{i,i-m} is its element m - 1
        double d = bi[0].x - x;
        for (int m = 1; m <= W; ++m) {
            const int s = (i + W - m) % W;
            d -= norm2(win[(s * W + m - 1) * stride]) * rinv[s * stride];
        }
        if (fabs(d) < tiny) d = -tiny;
        cnt += d < 0.0;
        cplx col[W];
#pragma unroll
        for (int k = 1; k <= W; ++k) col[k - 1] = (i + k < n) ? bi[k] : mk(0.0, 0.0);
        for (int m = 1; m < W; ++m) {
            const int s = (i + W - m) % W;
            const cplx piv = win[(s * W + m - 1) * stride] * rinv[s * stride];          // M_{i,i-m} / d_{i-m}
#pragma unroll
            for (int k = 1; k <= W; ++k)
                if (m + k <= W) col[k - 1] = col[k - 1] - mulc(win[(s * W + m + k - 1) * stride], piv);
        }
        if (d_out) {
            d_out[i] = d;
            for (int k = 0; k < W; ++k) m_out[(size_t)i * W + k] = col[k];
        }
        const int s = i % W;
        rinv[s * stride] = 1.0 / d;
#pragma unroll
        for (int k = 0; k < W; ++k) win[(s * W + k) * stride] = col[k];
    }
    return cnt;
}

template <int W>
__device__ __forceinline__ double bq_multisect(const cplx* band, int n, int target, double lo, double hi, double tiny, int lane,
                                      cplx* win, double* rinv) {
    for (int round = 0; round < 48; ++round) {
        const double wdt = hi - lo;
        if (!(wdt > 0.0)) break;
        const double x = lo + wdt * ((double)(lane + 1) / 65.0);
        const int ok = (x > lo && x < hi) ? (bq_band_count<W>(band, n, x, tiny, win + lane, rinv + lane, 64) >= target) : 0;
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

// dynamic LDS of the check kernel (bytes): band | sv | fd | window (64 lanes; later the factor M of the
// inverse iteration) | rinv | lane-0 window | packed A, B and B upper of the last step
template <int W> struct QCheckLds {
    static constexpr int NMAX = W * kq_max_steps<W>(), S = W * W;
    static constexpr size_t band = 0;
    static constexpr size_t sv = band + sizeof(cplx) * NMAX * (W + 1);
    static constexpr size_t fd = sv + sizeof(cplx) * NMAX;
    static constexpr size_t win = fd + sizeof(double) * NMAX;
    static constexpr size_t win_bytes = sizeof(cplx) * 64 * W * W > sizeof(cplx) * NMAX * W ? sizeof(cplx) * 64 * W * W
                                                                                           : sizeof(cplx) * NMAX * W;
    static constexpr size_t rinv = win + win_bytes;
    static constexpr size_t win0 = rinv + sizeof(double) * 64 * W;
    static constexpr size_t rinv0 = win0 + sizeof(cplx) * W * W;
    static constexpr size_t last = rinv0 + sizeof(double) * W;          // packed A | packed B | B upper (complex)
    static constexpr size_t total = last + sizeof(double) * (2 * S + 2 * S);
    static_assert(total <= 160 * 1024, "the check kernel's LDS must fit one CU");
};

template <int W>
__global__ void __launch_bounds__(64) pkq_check_kernel(const PackedJob* jobs, int launches_done) {
    constexpr int S = W * W;
    typedef QCoef<W> C;
    typedef QCheckLds<W> L;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* band = (cplx*)(smem_raw + L::band);
    cplx* sv = (cplx*)(smem_raw + L::sv);
    double* fd = (double*)(smem_raw + L::fd);
    cplx* win = (cplx*)(smem_raw + L::win);
    cplx* fm = win;                                 // the factor of the inverse iteration reuses the windows
    double* rinv = (double*)(smem_raw + L::rinv);
    cplx* win0 = (cplx*)(smem_raw + L::win0);
    double* rinv0 = (double*)(smem_raw + L::rinv0);
    double* lastA = (double*)(smem_raw + L::last);
    double* lastB = lastA + S;
    cplx* lastBu = (cplx*)(lastB + S);
    const PackedJob jb = jobs[blockIdx.x];
    if (jb.gen <= 0 || jb.state[0] >= jb.gen) return;      // idle slot / finished job
    const int lane = threadIdx.x;
    const int k_done = launches_done - jb.start;           // block steps this job has completed
    if (jb.n < 2) {
        if (lane == 0) {
            jb.state[0] = jb.gen;
            jb.status_out[0] = SCINT_E_EMPTY;
            jb.eig_out[0] = nan("");
            if (jb.iters_out) jb.iters_out[0] = 0;
        }
        return;
    }
    if (k_done < 2 && k_done < jb.max_steps) return;
    const int k_run = min(k_done, jb.max_steps);
    // blocks that make the Krylov space complete (all of them unless the space is saturated)
    __shared__ int rk[kq_max_steps<W>()];
    int rank = 0;
    const int k = bw_complete_steps_wave<W>(jb.beta, k_run, jb.n, lane, rk, &rank);
    const bool complete = rank >= jb.n;
    const int n = W * k;
    // A_{k-1}, B_{k-1}: pkq_coef_kernel ran for step k_run just before this kernel; when the space was
    // complete before the last step, the diagonal block comes from the history instead
    for (int i = lane; i < S; i += 64) {
        lastA[i] = k < k_run ? jb.alpha[S * (k - 1) + i] : gload(jb.coef + C::a_pack + i);
        lastB[i] = gload(jb.coef + C::b_pack + i);
        lastBu[i] = mk(gload(jb.coef + C::b_up + 2 * i), gload(jb.coef + C::b_up + 2 * i + 1));
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const int j = i / W, r = i - j * W;
        const double* A = j < k - 1 ? jb.alpha + S * j : lastA;
        const double* B = j + 1 < k ? jb.beta + S * (j + 1) : nullptr;      // couples blocks j and j + 1
        for (int kk = 0; kk <= W; ++kk) band[i * (W + 1) + kk] = bw_band_entry<W>(A, B, r, kk);
    }
    __syncthreads();
    double lo = INFINITY, hi = -INFINITY, scale = 0.0;
    for (int i = lane; i < n; i += 64) {
        double off = 0.0;
        for (int kk = 1; kk <= W; ++kk) {
            if (i + kk < n) off += sqrt(norm2(band[i * (W + 1) + kk]));
            if (i - kk >= 0) off += sqrt(norm2(band[(i - kk) * (W + 1) + kk]));
        }
        const double dg = band[i * (W + 1)].x;
        lo = fmin(lo, dg - off);
        hi = fmax(hi, dg + off);
        scale = fmax(scale, fabs(dg) + off);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, 64));
        hi = fmax(hi, __shfl_xor(hi, o, 64));
        scale = fmax(scale, __shfl_xor(scale, o, 64));
    }
    double bn2 = 0.0;
    for (int i = 0; i < S; ++i) bn2 += norm2(lastBu[i]);
    const double bnorm = sqrt(bn2);
    const bool finite = isfinite(lo) && isfinite(hi) && isfinite(bnorm);
    double theta = nan(""), theta2 = -INFINITY, resid = nan(""), err = nan("");
    if (finite && scale == 0.0 && bnorm == 0.0) {
        theta = 0.0; theta2 = 0.0; resid = 0.0; err = 0.0;     // all-zero theta-theta
    } else if (finite) {
        const double tiny = scale * 1e-300 + 1e-300;
        lo = lo - 1e-15 * fabs(lo) - 1e-300;
        hi = hi + 1e-15 * fabs(hi) + 1e-300;
        theta = bq_multisect<W>(band, n, n, lo, hi, tiny, lane, win, rinv);
        if (n >= 2) theta2 = bq_multisect<W>(band, n, n - 1, lo, theta, tiny, lane, win, rinv);
        __syncthreads();                            // every lane is done with its window before it becomes `fm`
        if (lane == 0) {
            const double sigma = theta + 8e-16 * fmax(fabs(theta), scale * 1e-3);
            bq_band_count<W>(band, n, sigma, tiny, win0, rinv0, 1, fd, fm);
            const double nrm = bw_inverse_iteration<W>(fd, fm, n, sv);
            double r2 = 0.0;                        // || B_{k-1} s_last ||^2
            for (int r = 0; r < W; ++r) {
                cplx v = mk(0.0, 0.0);
                for (int c = r; c < W; ++c) v = v + lastBu[r * W + c] * sv[n - W + c];
                r2 += norm2(v);
            }
            resid = nrm > 0.0 ? sqrt(r2 / nrm) : bnorm;
            if (!isfinite(resid)) resid = bnorm;
            if (jb.want_vec) {
                cplx* out = (cplx*)jb.svec;
                const double inv = nrm > 0.0 ? 1.0 / sqrt(nrm) : 0.0;
                for (int i = 0; i < n; ++i) out[i] = sv[i] * inv;
            }
        }
        resid = __shfl(resid, 0, 64);
        const double gap = theta - theta2;
        err = (gap > resid) ? resid * resid / gap : resid;
    }
    if (lane == 0) {
        const double prev = jb.result[3];
        const double at = fmax(fabs(theta), 1e-300);
        const bool settled = (theta - prev) <= 1e3 * jb.tol * at;
        // the projection is the whole matrix (accumulated rank, bw_complete_steps) or the last block
        // is exhausted (invariant subspace)
        const bool exact = finite && (complete || bnorm == 0.0);
        const double prev2 = jb.result[1], gap2 = theta - theta2;
        const bool gap_ok = gap2 > 0.0 && fabs(theta2 - prev2) <= 0.02 * gap2;
        const bool vec_ok = (resid <= jb.tol * at) || (gap_ok && settled && resid <= 30.0 * jb.tol * gap2);
        const bool ok = jb.want_vec ? vec_ok : (err <= jb.tol * at && settled);
        const bool conv = finite && (ok || exact);
        const bool stop = conv || !finite || k_run >= jb.max_steps;
        jb.result[0] = theta; jb.result[1] = theta2; jb.result[2] = resid; jb.result[3] = theta;
        if (stop) {
            jb.state[1] = k;
            jb.state[0] = jb.gen;
            jb.eig_out[0] = jb.want_vec ? theta : fabs(theta);   // modeler keeps the sign of w
            if (jb.iters_out) jb.iters_out[0] = k;
            jb.status_out[0] = (!finite || !isfinite(theta)) ? SCINT_E_NONFINITE : (conv ? SCINT_OK : SCINT_E_NOCONV);
        }
    }
}

}  // namespace scint
