// Stand-alone probe: which part of pk2_matvec_kernel (scintools_amd/csrc/eigen_packed.hip) costs
// the streaming rate?  tools/probes/stream_probe.hip reaches 6.7-7.0 TB/s with the kernel's load
// structure and no arithmetic; the product kernel measures 4.8 TB/s.  This file is the kernel body with
// its pieces switchable at compile time, on synthetic strips (every workgroup: 16 contiguous 64-KiB
// tiles of a 3 GiB buffer, X_J / X_I from a vector buffer):
//   bit 0  row part (acc += a x_J for 16 rows x 2 vectors)
//   bit 1  column part (c += conj(a) x_I, x_I by v_readlane)
//   bit 2  column partials through LDS every 4 tiles (two barriers) and stored
//   bit 3  prefetch of the next tile unconditional (else under `if (t + 1 < ntile)`)
//   bit 4  LDS-only barrier instead of __syncthreads() in the flush
//   bit 5  plain instead of non-temporal tile loads
//   bit 6  flush without its global stores (barriers + LDS traffic only)
//   bit 7  no LDS / barriers: every wave stores its own 16-row column partial each tile (4x the bytes)
//   bit 8  no LDS / barriers: wave 0 alone stores its partial each tile (the product's bytes, wrong sums)
//   bit 9  flush every 8 tiles instead of 4 (64 KiB of partials; X_J then comes from global memory)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk2_probe.hip -o /tmp/pk2_probe && /tmp/pk2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v2d __attribute__((ext_vector_type(2)));
#define GLOBAL __attribute__((address_space(1)))
struct __attribute__((aligned(16))) cplx { double x, y; };
__device__ inline cplx mk(double x, double y) { cplx r; r.x = x; r.y = y; return r; }
__device__ inline cplx operator+(cplx a, cplx b) { return mk(a.x + b.x, a.y + b.y); }
__device__ inline cplx operator*(cplx a, cplx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <bool NT> __device__ inline cplx gl(const cplx* p) {
    const v2d v = NT ? __builtin_nontemporal_load((const GLOBAL v2d*)p) : *(const GLOBAL v2d*)p;
    return mk(v.x, v.y);
}
__device__ inline void gs(cplx* p, cplx v) { v2d t; t.x = v.x; t.y = v.y; *(GLOBAL v2d*)p = t; }
__device__ inline double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kTB = 64, kTileElems = 4096, kStrip = 16;

template <int F>
__global__ void __launch_bounds__(256, 2)
probe_kernel(const cplx* __restrict__ tiles, const cplx* __restrict__ vec, cplx* __restrict__ colpart, cplx* __restrict__ rowpart, int ntile) {
    constexpr bool ROW = F & 1, COL = F & 2, FLUSH = F & 4, UNCOND = F & 8, LDSB = F & 16, NT = !(F & 32);
    constexpr bool NOST = F & 64, ALLW = F & 128, W0 = F & 256, F8 = F & 512;
    constexpr int kFlushF = F8 ? 8 : 4;
    __shared__ cplx cred[4][kFlushF][kTB][2];
    __shared__ cplx xs[F8 ? 1 : kStrip][kTB][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const cplx* __restrict__ tp = tiles + (size_t)blockIdx.x * ntile * kTileElems + (16 * w) * kTB + lane;
    cplx a0[8], a1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a0[r] = gl<NT>(tp + r * kTB);
    const cplx xI1 = gl<false>(vec + 2 * lane), xI2 = gl<false>(vec + 2 * lane + 1);
    if (!F8) for (int idx = threadIdx.x; idx < ntile * kTB; idx += 256) {
        xs[idx >> 6][idx & 63][0] = gl<false>(vec + 2 * idx);
        xs[idx >> 6][idx & 63][1] = gl<false>(vec + 2 * idx + 1);
    }
    if (LDSB) lds_barrier(); else __syncthreads();
    cplx acc1[16], acc2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = mk(0.0, 0.0); acc2[r] = mk(0.0, 0.0); }
    cplx sink = mk(0.0, 0.0);
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
        const cplx* __restrict__ tc = tp + (size_t)t * kTileElems;
#pragma unroll
        for (int r = 0; r < 8; ++r) a1[r] = gl<NT>(tc + (8 + r) * kTB);
        const cplx xJ1 = F8 ? gl<false>(vec + 2 * (t * kTB + lane)) : xs[t][lane][0], xJ2 = F8 ? gl<false>(vec + 2 * (t * kTB + lane) + 1) : xs[t][lane][1];
        cplx c1 = mk(0.0, 0.0), c2 = mk(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (ROW) { acc1[r] = acc1[r] + a0[r] * xJ1; acc2[r] = acc2[r] + a0[r] * xJ2; }
            if (COL) {
                const cplx x1 = mk(readlane_f64(xI1.x, 16 * w + r), readlane_f64(xI1.y, 16 * w + r));
                const cplx x2 = mk(readlane_f64(xI2.x, 16 * w + r), readlane_f64(xI2.y, 16 * w + r));
                c1 = mk(c1.x + a0[r].x * x1.x + a0[r].y * x1.y, c1.y + a0[r].x * x1.y - a0[r].y * x1.x);
                c2 = mk(c2.x + a0[r].x * x2.x + a0[r].y * x2.y, c2.y + a0[r].x * x2.y - a0[r].y * x2.x);
            }
            if (!ROW && !COL) sink = sink + a0[r];
        }
        if (UNCOND) {
            const cplx* __restrict__ nx = t + 1 < ntile ? tc + kTileElems : tc;
#pragma unroll
            for (int r = 0; r < 8; ++r) a0[r] = gl<NT>(nx + r * kTB);
        } else if (t + 1 < ntile) {
#pragma unroll
            for (int r = 0; r < 8; ++r) a0[r] = gl<NT>(tc + kTileElems + r * kTB);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (ROW) { acc1[8 + r] = acc1[8 + r] + a1[r] * xJ1; acc2[8 + r] = acc2[8 + r] + a1[r] * xJ2; }
            if (COL) {
                const cplx x1 = mk(readlane_f64(xI1.x, 16 * w + 8 + r), readlane_f64(xI1.y, 16 * w + 8 + r));
                const cplx x2 = mk(readlane_f64(xI2.x, 16 * w + 8 + r), readlane_f64(xI2.y, 16 * w + 8 + r));
                c1 = mk(c1.x + a1[r].x * x1.x + a1[r].y * x1.y, c1.y + a1[r].x * x1.y - a1[r].y * x1.x);
                c2 = mk(c2.x + a1[r].x * x2.x + a1[r].y * x2.y, c2.y + a1[r].x * x2.y - a1[r].y * x2.x);
            }
            if (!ROW && !COL) sink = sink + a1[r];
        }
        if (ALLW || W0) {
            if (ALLW || w == 0) {
                gs(colpart + 2 * ((((size_t)blockIdx.x * ntile + t) * (ALLW ? 4 : 1) + (ALLW ? w : 0)) * kTB + lane), c1);
                gs(colpart + 2 * ((((size_t)blockIdx.x * ntile + t) * (ALLW ? 4 : 1) + (ALLW ? w : 0)) * kTB + lane) + 1, c2);
            } else sink = sink + c1 + c2;
        } else if (FLUSH) {
            cred[w][t & (kFlushF - 1)][lane][0] = c1;
            cred[w][t & (kFlushF - 1)][lane][1] = c2;
            if ((t & (kFlushF - 1)) == kFlushF - 1 || t + 1 == ntile) {
                if (LDSB) lds_barrier(); else __syncthreads();
                const int tb = t & ~(kFlushF - 1);
#pragma unroll
                for (int c = w; c < 2 * kFlushF; c += 4) {
                    const int k = c >> 1, v = c & 1, tt = tb + k;
                    if (tt <= t) {
                        const cplx sum = ((cred[0][k][lane][v] + cred[1][k][lane][v]) + cred[2][k][lane][v]) + cred[3][k][lane][v];
                        if (NOST) sink = sink + sum; else gs(colpart + 2 * (((size_t)blockIdx.x * ntile + tt) * kTB + lane) + v, sum);
                    }
                }
                if (LDSB) lds_barrier(); else __syncthreads();
            }
        } else {
            sink = sink + c1 + c2;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double s1x = wave_sum(acc1[r].x), s1y = wave_sum(acc1[r].y), s2x = wave_sum(acc2[r].x), s2y = wave_sum(acc2[r].y);
        if (lane == 0) {
            gs(rowpart + 2 * ((size_t)blockIdx.x * kTB + 16 * w + r), mk(s1x, s1y));
            gs(rowpart + 2 * ((size_t)blockIdx.x * kTB + 16 * w + r) + 1, mk(s2x, s2y));
        }
    }
    if (sink.x == 1.2345e300) gs(rowpart, sink);
}

__global__ void fill_kernel(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
        p[i] = (double)(h & 0xFFFFF) * (1.0 / 1048576.0) - 0.5;
    }
}

static int g_reps = 5;
template <int F>
static void run(const cplx* buf, size_t bytes, const cplx* vec, cplx* colpart, cplx* rowpart, const char* what) {
    const int ntile = kStrip;
    const int nwg = (int)(bytes / ((size_t)ntile * 65536));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_kernel<F>, dim3(nwg), dim3(256), 0, 0, buf, vec, colpart, rowpart, ntile);
    hipEventRecord(e0, 0);
    const int reps = g_reps;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<F>, dim3(nwg), dim3(256), 0, 0, buf, vec, colpart, rowpart, ntile);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("flags %2d  %-58s %7.1f GB/s\n", F, what, (double)nwg * ntile * 65536.0 * reps / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    if (argc > 1) g_reps = atoi(argv[1]);   // sustained runs: 200 repetitions = 0.1 s per variant
    printf("repetitions per variant: %d\n", g_reps);
    const size_t bytes = (size_t)3 << 30;
    cplx *buf, *vec, *colpart, *rowpart;
    const size_t nwg = bytes / ((size_t)kStrip * 65536);
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&vec, 1 << 20) != hipSuccess ||
        hipMalloc(&colpart, nwg * kStrip * kTB * 2 * 16 * 4) != hipSuccess || hipMalloc(&rowpart, nwg * kTB * 2 * 16 + 64) != hipSuccess) {
        printf("alloc failed\n"); return 1;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (double*)buf, bytes / 8);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (double*)vec, (size_t)(1 << 20) / 8);
    hipDeviceSynchronize();
    run<0>(buf, bytes, vec, colpart, rowpart, "loads only (cond prefetch)");
    run<8>(buf, bytes, vec, colpart, rowpart, "loads only (uncond prefetch)");
    run<1>(buf, bytes, vec, colpart, rowpart, "row part");
    run<2>(buf, bytes, vec, colpart, rowpart, "column part, no flush");
    run<3>(buf, bytes, vec, colpart, rowpart, "row + column part, no flush");
    run<6>(buf, bytes, vec, colpart, rowpart, "column part + flush (__syncthreads)");
    run<7>(buf, bytes, vec, colpart, rowpart, "product kernel: row + column + flush (__syncthreads)");
    run<7 + 16>(buf, bytes, vec, colpart, rowpart, "row + column + flush (LDS-only barrier)");
    run<7 + 8>(buf, bytes, vec, colpart, rowpart, "row + column + flush, uncond prefetch");
    run<7 + 8 + 16>(buf, bytes, vec, colpart, rowpart, "SCINT_PK2_PREFETCH=1 form: uncond + LDS-only barrier");
    run<7 + 32>(buf, bytes, vec, colpart, rowpart, "product kernel with plain (not nt) tile loads");
    run<3 + 8>(buf, bytes, vec, colpart, rowpart, "row + column, no flush, uncond");
    run<7 + 64>(buf, bytes, vec, colpart, rowpart, "product kernel, flush without its global stores");
    run<3 + 128>(buf, bytes, vec, colpart, rowpart, "no LDS/barriers: every wave stores its partial (4x bytes)");
    run<3 + 256>(buf, bytes, vec, colpart, rowpart, "no LDS/barriers: wave 0 stores its partial each tile");
    run<7 + 512>(buf, bytes, vec, colpart, rowpart, "product kernel, flush every 8 tiles, X_J from global");
    return 0;
}
