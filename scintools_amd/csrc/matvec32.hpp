// matvec32.hpp -- the mat-vec of the ITERATION phase of the mixed-precision sweep (included by eigen_packed.hip,
// after pk2_colsum): the same two-vector product as pk2_matvec_kernel, with the tiles read as complex64.
//
// Everything but the tile elements is what it is in the complex128 kernel: vectors, accumulators, partial sums and
// their fixed summation orders are float64 (an element is widened once, exactly, when it is used), so the kernel is
// an exact block-Lanczos operator for the matrix A~ = fl32(A * scale) -- a Hermitian matrix 2^-24-close to A.  What
// the sweep returns is never an eigenvalue of A~: see "Mixed precision" in eigen_packed.hip.
//
// Shape.  A complex64 tile is 32 KiB, so a workgroup takes up to FOUR block rows I .. I+3 over its <= 14 column
// tiles (the tile bytes of two complex128 rows): the rows share the X_J blocks in LDS and ONE column partial per
// column tile, i.e. the partial-vector write traffic per tile byte stays what it is in the complex128 kernel (that
// traffic is what holds the kernel below the streaming rate: profiles/r03_pk2e_probe.txt).  Wave w owns the
// 16-column slice 16w .. 16w+15 of every tile, all 64 rows; lane l = 8 rg + cg reads row 8j + rg, columns
// 16w + 2cg and 16w + 2cg + 1 (ONE 16-byte load: a wave load is eight full 128-byte row segments) for j = 0..7 --
// eight loads hold a whole tile, and two tiles are in flight per wave (16 KiB, as in the complex128 kernel, where
// the sixteen loads are the two halves of one tile).  Column partials finish inside the wave (a reduce-scatter over
// the eight row groups), row partials by shuffles over the eight column pairs and one LDS step over the four waves.
// With twice the elements per byte the float64 arithmetic is no longer free (the complex128 kernel is ~30 % VALU-busy
// at its rate, this one would be ~60 %), so the loop is written for instruction count: products as fused
// multiply-add chains and the reduce-scatter bring a pair of tiles from 1052 to 814 instructions per wave (528 float64
// operations for the 512 the products need, 32 cross-lane exchanges instead of 96).
// The tile loop is branch-free: tiles go two at a time, the odd one out is peeled behind the loop.
// 256 threads, 72 KiB of LDS: two workgroups per CU, with room for the 8-KiB reduce blocks beside them.
#pragma once

namespace scint {

constexpr int kL32Xs = 0, kL32Col = kMaxStrip32 * kTB * 2, kL32Xi = 2 * kL32Col, kL32Rsum = kL32Xi + kRows32 * kTB * 2,
              kL32Elems = kL32Rsum + 4 * kTB * 2;   // complex128 elements
constexpr size_t kMatvec32LdsBytes = sizeof(cplx) * kL32Elems;
static_assert(kMatvec32LdsBytes <= 72 * 1024, "two complex64 mat-vec workgroups and a reduce block per CU");

__device__ __forceinline__ void pk32_load_tile(v4f (&a)[8], const c32* __restrict__ p) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = gload_nt4(p + (8 * j) * kTB);     // row 8j + rg, columns col0, col0 + 1
}

// one tile: element (j, cc) = row 8j + rg, column col0 + cc
__device__ __forceinline__ void pk32_tile(const v4f (&a)[8], const cplx (*__restrict__ xir)[2], const cplx (&xJ1)[2],
                                          const cplx (&xJ2)[2], cplx (&acc1)[8], cplx (&acc2)[8], cplx (&c1)[2], cplx (&c2)[2]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const cplx x1 = xir[8 * j][0], x2 = xir[8 * j][1];               // row 8j + rg of X_I
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const cplx e = mk((double)(cc ? a[j].z : a[j].x), (double)(cc ? a[j].w : a[j].y));
            // (four fused multiply-adds per product: as `acc + e * x` the compiler forms the product first -- mul, fma,
            // add per component, 128 more float64 instructions per pair of tiles)
            acc1[j].x = fma(-e.y, xJ1[cc].y, fma(e.x, xJ1[cc].x, acc1[j].x));
            acc1[j].y = fma(e.y, xJ1[cc].x, fma(e.x, xJ1[cc].y, acc1[j].y));
            acc2[j].x = fma(-e.y, xJ2[cc].y, fma(e.x, xJ2[cc].x, acc2[j].x));
            acc2[j].y = fma(e.y, xJ2[cc].x, fma(e.x, xJ2[cc].y, acc2[j].y));
            c1[cc] = mk(c1[cc].x + e.x * x1.x + e.y * x1.y, c1[cc].y + e.x * x1.y - e.y * x1.x);   // conj(a) x_I
            c2[cc] = mk(c2[cc].x + e.x * x2.x + e.y * x2.y, c2[cc].y + e.x * x2.y - e.y * x2.x);
        }
    }
}

// Column partial of a tile over the eight row groups as a reduce-scatter: at every exchange a lane keeps the half it
// will own -- vector rg & 1 after the exchange with row group rg ^ 1 (lanes l ^ 8), chunk (rg >> 1) & 1 after rg ^ 2
// (l ^ 16) -- and the last exchange (l ^ 32) completes the sum: 8 double exchanges instead of the 24 of an all-reduce
// followed by a select (pk2_colsum).  Fixed order; the lane ends with the value of its slot (row groups 4..7 with the
// same values as 0..3), as pk2_colsum returns it.
__device__ __forceinline__ cplx pk32_colsum(const cplx (&c1)[2], const cplx (&c2)[2], int rg) {
    const bool v1 = rg & 1, ch1 = rg & 2;
    auto sel = [](bool p, cplx a, cplx b) { return mk(p ? a.x : b.x, p ? a.y : b.y); };
    auto xchg = [](cplx v, int o) { return mk(__shfl_xor(v.x, o, 64), __shfl_xor(v.y, o, 64)); };
    cplx m0 = sel(v1, c2[0], c1[0]), m1 = sel(v1, c2[1], c1[1]);          // the vector this lane keeps, both chunks
    const cplx s0 = sel(v1, c1[0], c2[0]), s1 = sel(v1, c1[1], c2[1]);    // the other vector goes to the partner
    m0 = m0 + xchg(s0, 8);
    m1 = m1 + xchg(s1, 8);
    cplx m = sel(ch1, m1, m0);
    const cplx s = sel(ch1, m0, m1);
    m = m + xchg(s, 16);
    return m + xchg(m, 32);
}

// One block row of the strip: tiles t = t0 .. ntile-1 (t0 < ntile) at tp + (t - t0) tiles; a0 holds tile t0.
// ADD: the column partials are added to what the rows above left (the tile tskip -- this row's diagonal tile -- adds
// nothing); else they are stored.  Then the row partials.
template <bool ADD>
__device__ __forceinline__ void pk32_row(const c32* __restrict__ tp, v4f (&a0)[8], int t0, int ntile, int tskip,
                                        const cplx (*__restrict__ xs)[kTB][2], const cplx (*__restrict__ xir)[2],
                                        cplx* __restrict__ cslot, cplx* __restrict__ scratch, cplx (*__restrict__ rsum)[kTB][2],
                                        cplx* __restrict__ rowpart, int w, int col0, int cg, int rg) {
    v4f a1[8];
    cplx acc1[8], acc2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc1[j] = mk(0.0, 0.0); acc2[j] = mk(0.0, 0.0); }
    auto tile_step = [&](const v4f (&a)[8], int t) {
        cplx xJ1[2], xJ2[2], c1[2], c2[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            xJ1[cc] = xs[t][col0 + cc][0]; xJ2[cc] = xs[t][col0 + cc][1];
            c1[cc] = mk(0.0, 0.0); c2[cc] = mk(0.0, 0.0);
        }
        pk32_tile(a, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        const cplx c = pk32_colsum(c1, c2, rg);
        if (!ADD) cslot[2 * (t * kTB)] = c;
        else {
            // one read-modify-write per slot: row groups 4..7 (same values as 0..3) go to a scratch element each
            // (an address select, not a branch: see pk2_row)
            cplx* __restrict__ dst = rg < 4 ? cslot + 2 * (t * kTB) : scratch;
            const cplx o = *dst;
            const double keep = t == tskip ? 0.0 : 1.0;
            *dst = mk(o.x + keep * c.x, o.y + keep * c.y);
        }
    };
    int t = t0;
#pragma unroll 1
    for (; t + 2 < ntile; t += 2) {                                       // tiles t and t+1; tile t+2 exists
        const c32* __restrict__ tc = tp + (int64_t)(t - t0) * kTileElems;
        // (fences on both sides of both prefetches: left alone, the compiler hoists the first conversions of a0 above
        // these loads and then waits for ALL of a0 before the first product -- vmcnt(1), loads, vmcnt(8) in the ISA)
        __builtin_amdgcn_sched_barrier(0);
        pk32_load_tile(a1, tc + kTileElems);
        __builtin_amdgcn_sched_barrier(0);
        tile_step(a0, t);
        __builtin_amdgcn_sched_barrier(0);
        pk32_load_tile(a0, tc + 2 * kTileElems);
        __builtin_amdgcn_sched_barrier(0);
        tile_step(a1, t + 1);
    }
    if (t + 2 == ntile) {                                                 // two tiles left: nothing to prefetch behind them
        pk32_load_tile(a1, tp + (int64_t)(t + 1 - t0) * kTileElems);
        tile_step(a0, t);
        tile_step(a1, t + 1);
    } else {
        tile_step(a0, t);                                                 // one tile left
    }
    // row partials: the 8 lanes of a row group (xor 1, 2, 4; fixed order), then the four waves (column slices)
    if (ADD) __syncthreads();                    // the previous row's totals have been read by everybody
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cplx s = v ? acc2[j] : acc1[j];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) s = mk(s.x + __shfl_xor(s.x, o, 64), s.y + __shfl_xor(s.y, o, 64));
            if (cg == 0) rsum[w][8 * j + rg][v] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kTB) {
        const int row = threadIdx.x >> 1, v = threadIdx.x & 1;
        const cplx tot = ((rsum[0][row][v] + rsum[1][row][v]) + rsum[2][row][v]) + rsum[3][row][v];
        gstore(rowpart + threadIdx.x, tot);
    }
}

__device__ __forceinline__ void pk32_matvec_body(const Strip32* __restrict__ sp, int launch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];             // kMatvec32LdsBytes = 72 KiB
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    cplx (*xs)[kTB][2] = reinterpret_cast<cplx (*)[kTB][2]>(lds + kL32Xs);     // [kMaxStrip32]: the blocks X_J = rows of Q_j
    cplx (*rsum)[kTB][2] = reinterpret_cast<cplx (*)[kTB][2]>(lds + kL32Rsum); // [4 waves][64 rows][2]
    const int step = launch - sp->start;
    if (step < 0 || step >= sp->max_steps) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3, col0 = 16 * w + 2 * cg;
    const int ntile = sp->ntile;
    const int lane_off = rg * kTB + col0;                                       // row rg, first column of the lane
    v4f a0[8];
    pk32_load_tile(a0, sp->tiles[0] + lane_off);
    const int32_t done = gload(sp->state);
    const cplx* __restrict__ X = sp->Q + (int64_t)(step % sp->qslots) * sp->qstride * 2;   // Q_j
    const int I = sp->I, J0 = sp->J0, nrows = sp->nrows;
    // X_I of every row of the group and the strip's X_J blocks: contiguous copies of rows of Q_j
    for (int idx = threadIdx.x; idx < nrows * 2 * kTB; idx += 256) lds[kL32Xi + idx] = gload(X + 2 * I * kTB + idx);
    for (int idx = threadIdx.x; idx < ntile * 2 * kTB; idx += 256) lds[kL32Xs + idx] = gload(X + 2 * J0 * kTB + idx);
    if (done >= sp->gen) return;                 // finished job (workgroup-uniform): its loads were harmless
    __syncthreads();
    const cplx (*__restrict__ xi)[2] = reinterpret_cast<const cplx (*)[2]>(lds + kL32Xi);
    // this lane's slot in a tile's [64][2] column partial: column col0 + (rg >> 1 & 1), vector rg & 1
    cplx* __restrict__ cslot = lds + kL32Col + 2 * (col0 + ((rg >> 1) & 1)) + (rg & 1);
    pk32_row<false>(sp->tiles[0] + lane_off, a0, 0, ntile, -1, xs, xi + rg, cslot, nullptr, rsum, sp->rowpart[0], w, col0, cg, rg);
#pragma unroll 1
    for (int r = 1; r < nrows; ++r) {
        // block row I + r over the same columns: its tiles start at column max(J0, I + r); its diagonal tile adds no
        // column partial (the column part of a diagonal tile is its row part)
        const int t0 = I + r > J0 ? I + r - J0 : 0;
        if (t0 < ntile) {
            const c32* __restrict__ tpr = sp->tiles[r] + lane_off;
            pk32_load_tile(a0, tpr);
            // (scratch elements of the upper row groups: the first row's X_I block, dead since that row's barrier)
            pk32_row<true>(tpr, a0, t0, ntile, I + r - J0, xs, xi + r * kTB + rg, cslot, lds + kL32Xi + 32 * w + (lane & 31), rsum,
                           sp->rowpart[r], w, col0, cg, rg);
        } else if (threadIdx.x < 2 * kTB) {
            gstore(sp->rowpart[r] + threadIdx.x, mk(0.0, 0.0));   // short strips (tests): the row has nothing in this column range
        }
    }
    // the strip's column partials in one burst (the slot of a diagonal tile is written too; nobody reads it)
    __syncthreads();
    cplx* __restrict__ colpart = sp->colpart;
    for (int idx = threadIdx.x; idx < ntile * 2 * kTB; idx += 256) gstore_nt(colpart + idx, lds[kL32Col + idx]);
}

__global__ void __launch_bounds__(256, 2)
pk2_matvec32_kernel(const Strip32* __restrict__ strips, int launch) {
    pk32_matvec_body(strips + blockIdx.x, launch);
}

// Complex64 strips (workgroups 0 .. n32-1) and the complex128 strips of the certificate passes in one launch: the
// few certificates of a pass fill the tail of the launch instead of running as a small launch of their own.
// Dynamic LDS: the larger of the two carves (76 KiB).
constexpr size_t kMatvecMixedLdsBytes = kMatvecLdsBytes > kMatvec32LdsBytes ? kMatvecLdsBytes : kMatvec32LdsBytes;
__global__ void __launch_bounds__(256, 2)
pk2_matvec_mixed_kernel(const Strip32* __restrict__ strips32, int n32, const Strip* __restrict__ strips64, int launch) {
    if ((int)blockIdx.x < n32) pk32_matvec_body(strips32 + blockIdx.x, launch);
    else pk2_matvec_body(strips64 + ((int)blockIdx.x - n32), launch);
}

}  // namespace scint
