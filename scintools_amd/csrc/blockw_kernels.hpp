// blockw_kernels.hpp -- kernels of the W-vector block Lanczos recurrence (W = 4), the wider sibling of
// the pk2_* kernels in eigen_packed.hip.  Included by eigen_packed.hip after its device helpers.
//
// STATUS: opt-in, selected only by SCINT_LANCZOS_BLOCK=4 (mat-vec form: SCINT_MATVEC_MFMA=0 vector
// FMAs, 1 matrix cores; 2 selects the kernel family of blockq_kernels.hpp instead).  Green on the host
// interpreter (tests/emu); THESE KERNELS HAVE NOT RUN ON A GPU YET.  The traffic model
// (tools/models/wide_block_traffic.py) predicts that the vector-FMA quarter-strip form below LOSES
// against the default (1.12x the sweep cost): its column partials are per quarter tile.
//
// Why: the sweep is bound by streaming the packed matrix once per pass; the passes per curvature
// fall with the block width (CPU model at N = 4095: 31.9 passes for W = 2, 24.6 for W = 4), while
// the arithmetic per 16-byte element (8 W real FMAs) stays far below the FMA rate for W = 4.
//
// Shape of the mat-vec: the row accumulators are what limits the width -- rows x W complex values per
// lane.  With 4 rows per wave they take 64 registers for W = 4, so a workgroup (4 waves) covers 16 rows
// of a tile and FOUR workgroups ("quarters") share a strip of <= 8 tiles; column partials come per
// quarter tile and the reduce kernel adds the four.
#pragma once
#include "blockw.hpp"

namespace scint {

constexpr int kStripW = 8;          // strip cap of the W-vector mat-vec (its X_J blocks live in LDS)
constexpr int kQuarters = 4;        // workgroups per strip, 16 tile rows each
constexpr int kMaxKW = 64;          // block steps held in LDS by the W-vector check kernel (T up to 256 x 256 for W = 4)
constexpr int kRedGroupsW = 8;      // wavefronts per reduce block (its LDS buffer is groups x 64 x W complex)

// start block: rows n/2, n/2 + 7, n/2 - 7, n/2 + 14, ... of theta-theta (neighbouring rows in a tiny matrix)
template <int W>
__device__ inline int bw_start_row(int n, int v) {
    const int off = (v & 1) ? 7 * ((v + 1) / 2) : -7 * (v / 2);
    const int lo = n / 2 - 7 * (W / 2), hi = n / 2 + 7 * ((W + 1) / 2);
    if (lo >= 0 && hi < n) return n / 2 + off;
    return (n / 2 + v) % max(n, 1);
}

// `beta`, `step`, `n`: the history of the B factors (their nonzero pivots are the ranks of the blocks
// 0 .. step-1) and the matrix size -- the room left in the Krylov space for the block being formed
template <int W>
__device__ inline BlkW<W> bw_step_wave(const double* __restrict__ ap, const double* __restrict__ up, int nb, int lane,
                                       const double* __restrict__ beta, int step, int n) {
    constexpr int S = W * W;
    double sa[S], sg[S];
#pragma unroll
    for (int c = 0; c < S; ++c) { sa[c] = 0.0; sg[c] = 0.0; }
    for (int i = lane; i < nb; i += 64) {
#pragma unroll
        for (int c = 0; c < S; ++c) { sa[c] += gload(ap + S * i + c); sg[c] += gload(up + S * i + c); }
    }
#pragma unroll
    for (int c = 0; c < S; ++c) { sa[c] = wave_sum(sa[c]); sg[c] = wave_sum(sg[c]); }
    // nonzero pivots of the earlier blocks, counted by the lanes in parallel (a serial loop over
    // step * W dependent loads would cost their latencies one after the other)
    double used = 0.0;
    for (int idx = lane; idx < step * W; idx += 64) used += gload(beta + S * (idx / W) + idx % W) > 0.0 ? 1.0 : 0.0;
    const int room = n - (int)wave_sum(used);
    return bw_from_sums<W>(sa, sg, room);
}

// the coefficients are wave-uniform: keep them in scalar registers
template <int W>
__device__ inline BlkW<W> bw_uniform(BlkW<W> k) {
#pragma unroll
    for (int r = 0; r < W; ++r) {
        k.inv[r] = uniform_f64(k.inv[r]);
#pragma unroll
        for (int c = r; c < W; ++c) {
            k.a[r][c] = mk(uniform_f64(k.a[r][c].x), uniform_f64(k.a[r][c].y));
            k.b[r][c] = mk(uniform_f64(k.b[r][c].x), uniform_f64(k.b[r][c].y));
        }
    }
    return k;
}

// row r of Q_j (all W columns) from W_{j-1} (Up) and Q_{j-1} (Qp), both interleaved [row][W]
template <int W>
__device__ inline void bw_q_row_at(const BlkW<W>& sc, const cplx* __restrict__ Up, const cplx* __restrict__ Qp, int r,
                                   cplx (&x)[W]) {
    cplx u[W], q[W];
#pragma unroll
    for (int v = 0; v < W; ++v) { u[v] = gload(Up + W * r + v); q[v] = gload(Qp + W * r + v); }
    bw_q_row<W>(sc, u, q, x);
}

template <int W>
__global__ void __launch_bounds__(64) pkw_init_kernel(const PackedJob* jobs, const int32_t* slots) {
    constexpr int S = W * W;
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K == 0 && e == 0) { jb.state[1] = 0; jb.result[1] = -INFINITY; jb.result[3] = -INFINITY; }
    if (K >= jb.nb) return;
    const int r = K * kTB + e;
    cplx x[W];
#pragma unroll
    for (int v = 0; v < W; ++v) x[v] = (r < jb.n && jb.n >= 2) ? packed_at(jb, bw_start_row<W>(jb.n, v), r) : mk(0.0, 0.0);
    cplx* qm1 = jb.Q + (int64_t)(jb.qslots - 1) * jb.qstride * W;     // "Q_{-1}" = 0
#pragma unroll
    for (int v = 0; v < W; ++v) {
        jb.U[0][W * r + v] = x[v];
        jb.U[1][W * r + v] = mk(0.0, 0.0);
        qm1[W * r + v] = mk(0.0, 0.0);
        jb.Q[W * r + v] = mk(0.0, 0.0);
    }
    // Gram matrix of the start block (packed); A-part zero: step 0 orthonormalises the block
    double pg[S];
#pragma unroll
    for (int a = 0; a < W; ++a) {
        pg[a] = wave_sum(norm2(x[a]));
#pragma unroll
        for (int b = a + 1; b < W; ++b) {
            const cplx z = wave_sum(mulc(x[b], x[a]));                 // conj(x_a) x_b
            pg[bw_upper<W>(a, b)] = z.x; pg[bw_upper<W>(a, b) + 1] = z.y;
        }
    }
    if (e == 0) {
#pragma unroll
        for (int c = 0; c < S; ++c) {
            jb.apart[0][S * K + c] = 0.0; jb.apart[1][S * K + c] = 0.0; jb.upart[1][S * K + c] = 0.0;
            jb.upart[0][S * K + c] = pg[c];
        }
    }
}

// One workgroup per QUARTER strip: rows 16 q .. 16 q + 15 of the tiles (I, J0..J1); wave w owns rows
// 16 q + 4 w .. + 3 against all W vectors.  Tiles are double-buffered in registers (8 independent
// 1-KiB wave loads in flight per wave, 3 workgroups per CU).
template <int W>
__global__ void __launch_bounds__(256, 3)
pkw_matvec_kernel(const PackedJob* __restrict__ jobs, const Strip* __restrict__ strips, int launch) {
    __shared__ cplx cred[4][kTB][W];            // per-wave column partials of the current tile
    __shared__ cplx xs[kStripW][kTB][W];        // the blocks X_J of the strip, rebuilt once per workgroup
    const Strip st = strips[blockIdx.x >> 2];
    const int qr = blockIdx.x & 3;
    const PackedJob* __restrict__ jp = jobs + st.job;
    const int step = launch - jp->start;
    if (jp->n < 2 || step < 0 || step >= jp->max_steps || gload(jp->state) >= jp->gen) return;
    const int par = step & 1;
    const int nb = jp->nb;
    const cplx* __restrict__ Up = par ? jp->U[1] : jp->U[0];
    const int qs = jp->qslots;
    const cplx* __restrict__ Qp = jp->Q + (int64_t)((step + qs - 1) % qs) * jp->qstride * W;   // Q_{j-1}
    const cplx* __restrict__ tiles = jp->tiles;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int I = st.I;
    const int64_t t0 = tile_offset(nb, I);
    const int ntile = st.J1 - st.J0;
    const int row0 = 16 * qr + 4 * w;                      // this wave's 4 rows inside the tile
    const cplx* __restrict__ tp = tiles + (t0 + (st.J0 - I)) * kTileElems + row0 * kTB + lane;
    cplx a0[4], a1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = gload_nt(tp + r * kTB);
    const BlkW<W> sc = bw_uniform<W>(bw_step_wave<W>(par ? jp->apart[1] : jp->apart[0], par ? jp->upart[1] : jp->upart[0], nb, lane, jp->beta, step, jp->n));
    // lanes 0..15 of every wave hold rows 16 q .. 16 q + 15 of the block X_I; rows read them back with v_readlane
    cplx xI[W];
    bw_q_row_at<W>(sc, Up, Qp, I * kTB + 16 * qr + (lane & 15), xI);
    // X_J = rows J0*64 .. J1*64 of Q_j, once per workgroup (the first tile's loads stay in flight)
    for (int idx = threadIdx.x; idx < ntile * kTB; idx += 256) {
        cplx x[W];
        bw_q_row_at<W>(sc, Up, Qp, st.J0 * kTB + idx, x);
#pragma unroll
        for (int v = 0; v < W; ++v) xs[idx >> 6][idx & 63][v] = x[v];
    }
    __syncthreads();
    cplx acc[4][W];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < W; ++v) acc[r][v] = mk(0.0, 0.0);
    cplx* __restrict__ colpart = jp->colpart;
    // one tile: 4 rows x 64 columns of this wave against the W vectors
    auto tile_step = [&](const cplx (&a)[4], int t) {
        cplx xJ[W], c[W];
#pragma unroll
        for (int v = 0; v < W; ++v) { xJ[v] = xs[t][lane][v]; c[v] = mk(0.0, 0.0); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int v = 0; v < W; ++v) {
                acc[r][v] = acc[r][v] + a[r] * xJ[v];
                const cplx x = mk(readlane_f64(xI[v].x, 4 * w + r), readlane_f64(xI[v].y, 4 * w + r));
                c[v] = mk(c[v].x + a[r].x * x.x + a[r].y * x.y, c[v].y + a[r].x * x.y - a[r].y * x.x);   // conj(a) x_I
            }
        }
#pragma unroll
        for (int v = 0; v < W; ++v) cred[w][lane][v] = c[v];
        __syncthreads();
        const int Jt = st.J0 + t;
        if (Jt != I) {
            // cross-wave sum of the tile's column partials: wave w takes the vectors w, w + 4, ...
            for (int v = w; v < W; v += 4) {
                const cplx sum = ((cred[0][lane][v] + cred[1][lane][v]) + cred[2][lane][v]) + cred[3][lane][v];
                gstore(colpart + W * (((t0 + (Jt - I)) * kQuarters + qr) * kTB + lane) + v, sum);
            }
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int t = 0; t < ntile; t += 2) {
        const cplx* __restrict__ tc = tp + (int64_t)t * kTileElems;
        if (t + 1 < ntile) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[r] = gload_nt(tc + kTileElems + r * kTB);           // tile t + 1
        }
        tile_step(a0, t);
        if (t + 1 < ntile) {
            if (t + 2 < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[r] = gload_nt(tc + 2 * kTileElems + r * kTB);   // tile t + 2
            }
            tile_step(a1, t + 1);
        }
    }
    cplx* __restrict__ rowpart = jp->rowpart;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int v = 0; v < W; ++v) {
            const cplx s = wave_sum(acc[r][v]);
            if (lane == 0) gstore(rowpart + W * ((int64_t)st.index * kTB + row0 + r) + v, s);
        }
    }
}

// ---- matrix-core form of the W-vector mat-vec -------------------------------------------------
// STATUS: checked on the host interpreter only (tests/emu, incl. its model of v_mfma_f64_16x16x4_f64
// taken from the programming guide's operand layout); selected by SCINT_MATVEC_MFMA=1 together with
// SCINT_LANCZOS_BLOCK=4.  Never run on a GPU.
//
// Why: with vector FMAs the row partials sum_j a[r][j] x_J[j][v] live in rows x W complex accumulators
// PER LANE (lane = column j), which is what caps the block width (256 registers for 16 rows x 4
// vectors) or forces the quarter strips above (whose column partials cost 25 % extra traffic each
// way).  v_mfma_f64_16x16x4 contracts over the lanes instead: a 16-row x 64-column slab of a tile
// against the W vectors (2 W <= 16 real columns) accumulates in FOUR doubles per lane.  A complex
// product is two real ones, with X as the B operand:
//   rows:     D[r][n] += Re a[r][j] * X1[j][n] + Im a[r][j] * X2[j][n],   X1 = (xr, xi) interleaved, X2 = i X
//   columns:  D[j][n] += Re a[r][j] * Y1[r][n] + Im a[r][j] * Y3[r][n],   Y1 = (xr, xi),  Y3 = -i X_I  (conj(a) x)
// The matrix operand wants the contracted index in lane >> 4 and the free one in lane & 15, so a wave
// reads its 16 rows of a tile twice: as 16 x 4 patches (row part) and as 4 x 16 patches (column part);
// the second read hits L1/L2.  One workgroup (4 waves, 16 tile rows each) per strip of <= kStripW
// tiles; column partials are summed across the waves through LDS per tile -> ONE partial per tile.
// For W = 4 half of the 16 B-columns are zero (the same instruction count would carry W = 8).
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ inline v4d mfma_f64_16x16x4(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// Operand layout the kernels assume (cdna_hip_programming.md, "f64 MFMA"; the same follows from
// composable_kernel's descriptor of this instruction, xdlops_gemm.hpp mfma_type<mfma_f64_16x16x4f64>:
// group_size 1, 4 groups, 16 threads per block, 4 input blocks, k reduction): A[i][k] in lane 16 k + i,
// B[k][j] in lane 16 k + j, and register r of lane l of C/D holds row mfma_d_row(l, r), column l & 15.
// If a GPU run of the wide-block test fails only for the matrix-core forms, this is the one place
// (with its mirror in tests/emu/include/hip/hip_runtime.h) to look at first.
__device__ inline int mfma_d_row(int lane, int r) { return (lane >> 4) + 4 * r; }

template <int W>
__global__ void __launch_bounds__(256, 2)
pkw_matvec_mfma_kernel(const PackedJob* __restrict__ jobs, const Strip* __restrict__ strips, int launch) {
    constexpr int NR = 2 * W;                      // real B columns in use (of 16)
    static_assert(NR <= 16 && (NR & (NR - 1)) == 0, "block width must be 1, 2, 4 or 8");
    __shared__ double xs[kStripW][kTB][NR];        // the blocks X_J of the strip as real columns (re0, im0, re1, ...)
    __shared__ double xI[kTB][NR];                 // the block X_I
    __shared__ double cred[4][kTB][NR];            // per-wave column partials of the current tile
    const Strip st = strips[blockIdx.x];
    const PackedJob* __restrict__ jp = jobs + st.job;
    const int step = launch - jp->start;
    if (jp->n < 2 || step < 0 || step >= jp->max_steps || gload(jp->state) >= jp->gen) return;
    const int par = step & 1;
    const int nb = jp->nb;
    const cplx* __restrict__ Up = par ? jp->U[1] : jp->U[0];
    const int qs = jp->qslots;
    const cplx* __restrict__ Qp = jp->Q + (int64_t)((step + qs - 1) % qs) * jp->qstride * W;   // Q_{j-1}
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k4 = lane >> 4, n16 = lane & 15;     // contracted index / free index of the matrix-core operands
    const bool odd = n16 & 1;
    const double live = n16 < NR ? 1.0 : 0.0;      // B columns >= 2 W are zero
    const int npair = (n16 & (NR - 1)) & ~1;       // this lane's (re, im) pair in a row of xs
    const int I = st.I;
    const int64_t t0 = tile_offset(nb, I);
    const int ntile = st.J1 - st.J0;
    const cplx* __restrict__ tb = jp->tiles + (t0 + (st.J0 - I)) * kTileElems;
    // row-part operand g of a tile:    a[16 w + n16][4 g + k4]            g = 0..15
    // column-part operand (c, kk):     a[16 w + 4 kk + k4][16 c + n16]    c, kk = 0..3
    const cplx* __restrict__ rp = tb + (16 * w + n16) * kTB + k4;
    const cplx* __restrict__ cp = tb + (16 * w + k4) * kTB + n16;
    cplx ra[16], ca[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) ra[g] = gload(rp + 4 * g);
    const BlkW<W> sc = bw_uniform<W>(bw_step_wave<W>(par ? jp->apart[1] : jp->apart[0], par ? jp->upart[1] : jp->upart[0], nb, lane, jp->beta, step, jp->n));
    for (int idx = threadIdx.x; idx < (ntile + 1) * kTB; idx += 256) {
        const bool own = idx >= ntile * kTB;       // the last 64 entries build X_I
        cplx x[W];
        bw_q_row_at<W>(sc, Up, Qp, own ? I * kTB + (idx - ntile * kTB) : st.J0 * kTB + idx, x);
        double* dst = own ? &xI[idx - ntile * kTB][0] : &xs[idx >> 6][idx & 63][0];
#pragma unroll
        for (int v = 0; v < W; ++v) { dst[2 * v] = x[v].x; dst[2 * v + 1] = x[v].y; }
    }
    lds_barrier();
    // B operands of the column part, fixed for the strip: rows 16 w + 4 kk + k4 of X_I
    double y1[4], y3[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double pr = xI[16 * w + 4 * kk + k4][npair], pi = xI[16 * w + 4 * kk + k4][npair + 1];
        y1[kk] = live * (odd ? pi : pr);           // conj(a) x = (ar xr + ai xi) + i (ar xi - ai xr)
        y3[kk] = live * (odd ? -pr : pi);
    }
    v4d accr0 = {0.0, 0.0, 0.0, 0.0}, accr1 = {0.0, 0.0, 0.0, 0.0};
    double* __restrict__ colpart = (double*)jp->colpart;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const int64_t toff = (int64_t)t * kTileElems;
#pragma unroll
        for (int q = 0; q < 16; ++q) ca[q] = gload(cp + toff + (4 * (q & 3)) * kTB + 16 * (q >> 2));   // q = 4 c + kk
        // rows: 16 patches of 16 rows x 4 columns against X_J
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const double pr = xs[t][4 * g + k4][npair], pi = xs[t][4 * g + k4][npair + 1];
            const double x1 = live * (odd ? pi : pr);          // a x = (ar xr - ai xi) + i (ar xi + ai xr)
            const double x2 = live * (odd ? pr : -pi);
            if (g & 1) { accr1 = mfma_f64_16x16x4(ra[g].x, x1, accr1); accr1 = mfma_f64_16x16x4(ra[g].y, x2, accr1); }
            else       { accr0 = mfma_f64_16x16x4(ra[g].x, x1, accr0); accr0 = mfma_f64_16x16x4(ra[g].y, x2, accr0); }
        }
        {   // row patches of the next tile (of this one again after the last: an unconditional load keeps
            // the compiler's wait counts exact -- a load under a branch makes them pessimistic)
            const int64_t noff = t + 1 < ntile ? toff + kTileElems : toff;
#pragma unroll
            for (int g = 0; g < 16; ++g) ra[g] = gload(rp + noff + 4 * g);
        }
        // columns: for each block of 16 columns, 4 patches of 4 rows x 16 columns against X_I
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
            // cross-wave sum, fixed order; the tile's partial as [64 columns][W] complex
            for (int idx = threadIdx.x; idx < kTB * NR; idx += 256) {
                const int col = idx / NR, nn = idx - col * NR;
                const double sum = ((cred[0][col][nn] + cred[1][col][nn]) + cred[2][col][nn]) + cred[3][col][nn];
                gstore(colpart + NR * ((t0 + (Jt - I)) * kTB + col) + nn, sum);
            }
        }
        lds_barrier();
    }
    // row r = 16 w + k4 + 4 reg of the strip's row partial, real column n16
    double* __restrict__ rowpart = (double*)jp->rowpart;
    if (n16 < NR) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            gstore(rowpart + NR * ((int64_t)st.index * kTB + 16 * w + mfma_d_row(lane, r)) + n16, accr0[r] + accr1[r]);
    }
}

template <int W>
__global__ void __launch_bounds__(64 * kRedGroupsW)
pkw_reduce_kernel(const PackedJob* __restrict__ jobs, int launch, int cparts) {
    constexpr int S = W * W;
    __shared__ cplx part[kRedGroupsW][kTB][W];
    const PackedJob jb = jobs[blockIdx.y];
    const int K = blockIdx.x;
    const int step = launch - jb.start;
    if (K >= jb.nb || jb.n < 2 || step < 0 || step >= jb.max_steps || gload(jb.state) >= jb.gen) return;
    const int par = step & 1;
    const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
    // fixed summation order: the strips of block row K, then the column partials of the tiles
    // (cI, K), cI < K, part by part (cparts = kQuarters for the quarter-strip mat-vec, 1 for the
    // matrix-core one)
    const int s0 = jb.row_strip0[K], nrow = jb.row_strip0[K + 1] - s0;
    cplx acc[W];
#pragma unroll
    for (int v = 0; v < W; ++v) acc[v] = mk(0.0, 0.0);
    for (int idx = g; idx < nrow + cparts * K; idx += kRedGroupsW) {
        const int ci = idx - nrow;
        const int cI = ci / cparts, cq = ci - cI * cparts;
        const cplx* src = idx < nrow ? jb.rowpart + W * ((int64_t)(s0 + idx) * kTB + e)
                                     : jb.colpart + W * (((tile_offset(jb.nb, cI) + (K - cI)) * cparts + cq) * kTB + e);
#pragma unroll
        for (int v = 0; v < W; ++v) acc[v] = acc[v] + gload(src + v);
    }
#pragma unroll
    for (int v = 0; v < W; ++v) part[g][e][v] = acc[v];
    __syncthreads();
    if (g == 0) {
        const BlkW<W> sc = bw_step_wave<W>(par ? jb.apart[1] : jb.apart[0], par ? jb.upart[1] : jb.upart[0], jb.nb, e, jb.beta, step, jb.n);
        cplx tot[W];
#pragma unroll
        for (int v = 0; v < W; ++v) {
            tot[v] = part[0][e][v];
#pragma unroll
            for (int k = 1; k < kRedGroupsW; ++k) tot[v] = tot[v] + part[k][e][v];
        }
        const int r = K * kTB + e;
        const cplx* __restrict__ Up = par ? jb.U[1] : jb.U[0];
        const cplx* __restrict__ Qp = jb.Q + (int64_t)((step + jb.qslots - 1) % jb.qslots) * jb.qstride * W;
        cplx* __restrict__ Un = par ? jb.U[0] : jb.U[1];
        cplx* __restrict__ Qn = jb.Q + (int64_t)(step % jb.qslots) * jb.qstride * W;
        cplx u[W], q[W], x[W], qbh[W], t[W];
#pragma unroll
        for (int v = 0; v < W; ++v) { u[v] = gload(Up + W * r + v); q[v] = gload(Qp + W * r + v); }
        bw_q_row<W>(sc, u, q, x);                          // row of Q_j
        bw_qbh_row<W>(sc, q, qbh);                         // row of Q_{j-1} B_{j-1}^H
#pragma unroll
        for (int v = 0; v < W; ++v) {
            t[v] = tot[v] - qbh[v];                        // row of W_j = A Q_j - Q_{j-1} B_{j-1}^H
            gstore(Un + W * r + v, t[v]);
            gstore(Qn + W * r + v, x[v]);
        }
        double pa[S], pg[S];                               // packed partials of A_j = Q_j^H W_j and of W_j^H W_j
#pragma unroll
        for (int a = 0; a < W; ++a) {
            pa[a] = wave_sum(x[a].x * t[a].x + x[a].y * t[a].y);
            pg[a] = wave_sum(norm2(t[a]));
#pragma unroll
            for (int b = a + 1; b < W; ++b) {
                const cplx za = wave_sum(mulc(t[b], x[a]));            // conj(x_a) t_b
                const cplx zg = wave_sum(mulc(t[b], t[a]));            // conj(t_a) t_b
                pa[bw_upper<W>(a, b)] = za.x; pa[bw_upper<W>(a, b) + 1] = za.y;
                pg[bw_upper<W>(a, b)] = zg.x; pg[bw_upper<W>(a, b) + 1] = zg.y;
            }
        }
        if (e == 0) {
            double* an = par ? jb.apart[0] : jb.apart[1];
            double* un = par ? jb.upart[0] : jb.upart[1];
#pragma unroll
            for (int c = 0; c < S; ++c) { an[S * K + c] = pa[c]; un[S * K + c] = pg[c]; }
            if (K == 0) {
                double ca[S], cb[S];
                bw_pack<W>(sc, ca, cb);
                if (step > 0) {
#pragma unroll
                    for (int c = 0; c < S; ++c) jb.alpha[S * (step - 1) + c] = ca[c];
                }
#pragma unroll
                for (int c = 0; c < S; ++c) jb.beta[S * step + c] = cb[c];   // B[step] couples blocks step-1 and step
            }
        }
    }
}

// bw_complete_steps for one wavefront: the ranks of the blocks are read by the lanes in parallel into
// `rk` (LDS, >= k entries), then scanned (the serial version waits for k * W loads one after the other)
template <int W>
__device__ inline int bw_complete_steps_wave(const double* __restrict__ beta, int k, int n, int lane, int* rk, int* rank_out) {
    for (int j = lane; j < k; j += 64) {
        int c = 0;
#pragma unroll
        for (int r = 0; r < W; ++r) c += gload(beta + W * W * j + r) > 0.0;
        rk[j] = c;
    }
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < k; ++j) {
        rank += rk[j];
        if (rank >= n) { *rank_out = rank; return j + 1; }
    }
    *rank_out = rank;
    return k;
}

template <int W>
__device__ inline double bw_multisect(const cplx* band, int n, int target, double lo, double hi, double tiny, int lane) {
    for (int round = 0; round < 48; ++round) {
        const double wdt = hi - lo;
        if (!(wdt > 0.0)) break;
        const double x = lo + wdt * ((double)(lane + 1) / 65.0);
        const int ok = (x > lo && x < hi) ? (bw_band_count<W>(band, n, x, tiny) >= target) : 0;
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

template <int W>
__global__ void __launch_bounds__(64) pkw_check_kernel(const PackedJob* jobs, int launches_done) {
    constexpr int S = W * W;
    constexpr int NMAX = W * kMaxKW;
    __shared__ cplx band[NMAX * (W + 1)];
    __shared__ double fd[NMAX];                    // pivots of the factorisation used by the inverse iteration
    __shared__ cplx fm[NMAX * W];                  // its M_{i+k,i}
    __shared__ cplx sv[NMAX];
    __shared__ double lastA[S], lastB[S];
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
    // A_{k-1}, B_{k-1} are still in the partials of the last reduce kernel
    const BlkW<W> last = bw_step_wave<W>((k_run & 1) ? jb.apart[1] : jb.apart[0], (k_run & 1) ? jb.upart[1] : jb.upart[0], jb.nb, lane, jb.beta, k_run, jb.n);
    // blocks that make the Krylov space complete (all of them unless the space is saturated)
    __shared__ int rk[kMaxKW];
    int rank = 0;
    const int k = bw_complete_steps_wave<W>(jb.beta, k_run, jb.n, lane, rk, &rank);
    const bool complete = rank >= jb.n;
    const int n = W * k;
    if (lane == 0) {
        bw_pack<W>(last, lastA, lastB);
        if (k < k_run) {                       // saturated before the last step: its diagonal block is in the history
            for (int c = 0; c < S; ++c) lastA[c] = jb.alpha[S * (k - 1) + c];
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const int j = i / W, r = i - j * W;
        const double* A = j < k - 1 ? jb.alpha + S * j : lastA;
        const double* B = j + 1 < k ? jb.beta + S * (j + 1) : nullptr;      // couples blocks j and j + 1
#pragma unroll
        for (int kk = 0; kk <= W; ++kk) band[i * (W + 1) + kk] = bw_band_entry<W>(A, B, r, kk);
    }
    __syncthreads();
    double lo = INFINITY, hi = -INFINITY, scale = 0.0;
    for (int i = lane; i < n; i += 64) {
        double off = 0.0;
#pragma unroll
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
#pragma unroll
    for (int r = 0; r < W; ++r)
#pragma unroll
        for (int c = r; c < W; ++c) bn2 += norm2(last.b[r][c]);
    const double bnorm = sqrt(bn2);
    const bool finite = isfinite(lo) && isfinite(hi) && isfinite(bnorm);
    double theta = nan(""), theta2 = -INFINITY, resid = nan(""), err = nan("");
    if (finite && scale == 0.0 && bnorm == 0.0) {
        theta = 0.0; theta2 = 0.0; resid = 0.0; err = 0.0;     // all-zero theta-theta
    } else if (finite) {
        const double tiny = scale * 1e-300 + 1e-300;
        lo = lo - 1e-15 * fabs(lo) - 1e-300;
        hi = hi + 1e-15 * fabs(hi) + 1e-300;
        theta = bw_multisect<W>(band, n, n, lo, hi, tiny, lane);
        if (n >= 2) theta2 = bw_multisect<W>(band, n, n - 1, lo, theta, tiny, lane);
        if (lane == 0) {
            // Ritz vector by inverse iteration on T - sigma, sigma just above theta; only its last
            // block is needed for the residual: resid = || B_{k-1} s_last || / ||s||
            const double sigma = theta + 8e-16 * fmax(fabs(theta), scale * 1e-3);
            bw_band_count<W>(band, n, sigma, tiny, fd, fm);
            const double nrm = bw_inverse_iteration<W>(fd, fm, n, sv);
            resid = nrm > 0.0 ? sqrt(bw_resid2<W>(last, sv + (n - W)) / nrm) : bnorm;
            if (!isfinite(resid)) resid = bnorm;
            if (jb.want_vec) {                      // unit-norm eigenvector of T_k for the Ritz vector
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
        // eigenvector wanted: same gap-aware rule as the other check kernels
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

// Ritz vector of finished jobs: y = sum_j sum_v Q_j[:, v] s_{W j + v}; normalised by pk_ritz_scale_kernel
// (the partial norms go to the first nb entries of upart[0]).
template <int W>
__global__ void __launch_bounds__(64) pkw_ritz_kernel(const PackedJob* jobs, const int32_t* slots,
                                                      const int64_t* eta_index, cplx* vec_out, int64_t vstride) {
    const PackedJob jb = jobs[slots[blockIdx.y]];
    const int K = blockIdx.x, e = threadIdx.x;
    if (K >= jb.nb) return;
    const int k = jb.state[1];
    const int r = K * kTB + e;
    const cplx* __restrict__ sv = (const cplx*)jb.svec;
    cplx y = mk(0.0, 0.0);
    for (int j = 0; j < k; ++j) {
        const cplx* __restrict__ q = jb.Q + (int64_t)j * jb.qstride * W + W * r;
#pragma unroll
        for (int v = 0; v < W; ++v) y = y + q[v] * sv[W * j + v];
    }
    cplx* out = vec_out + eta_index[blockIdx.y] * vstride;
    if (r < jb.n) out[r] = y;
    const double p = wave_sum(r < jb.n ? norm2(y) : 0.0);
    if (e == 0) jb.upart[0][K] = p;   // the job is finished: its partial arrays are free
}

}  // namespace scint
