// Stand-alone probe of the round-3 mat-vec shape (scintools_amd/csrc/eigen_packed.hip, pk2_matvec_kernel):
// a wave owns all 64 rows of a 16-column slice, column partials finish inside the wave, no barrier in
// the tile loop.  Synthetic strips as in pk2_probe.hip (every workgroup: NT contiguous 64-KiB tiles of a
// 3 GiB buffer).  Variants:
//   bit 0  column partial store each tile (else accumulated into a sink)
//   bit 3  column partials into the consumed X_J slots in LDS, one coalesced burst at the end of the strip
//   bit 4  (with bit 3) the burst as non-temporal stores;  bit 5  (with bit 3) the burst into one 32-KiB region
//          shared by all workgroups (stays in L2: the cost of issuing the stores without their HBM traffic)
//   bit 1  three workgroups per CU (launch bound 3, 48 KiB LDS) instead of two
//   bit 2  x_J from registers re-read per half (shorter live ranges)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk2e_probe.hip -o /tmp/pk2e_probe && /tmp/pk2e_probe [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v2d __attribute__((ext_vector_type(2)));
#define GLOBAL __attribute__((address_space(1)))
struct __attribute__((aligned(16))) cplx { double x, y; };
__device__ inline cplx mk(double x, double y) { cplx r; r.x = x; r.y = y; return r; }
__device__ inline cplx operator+(cplx a, cplx b) { return mk(a.x + b.x, a.y + b.y); }
__device__ inline cplx operator*(cplx a, cplx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline cplx gl(const cplx* p) { const v2d v = *(const GLOBAL v2d*)p; return mk(v.x, v.y); }
__device__ inline cplx glnt(const cplx* p) { const v2d v = __builtin_nontemporal_load((const GLOBAL v2d*)p); return mk(v.x, v.y); }
__device__ inline void gs(cplx* p, cplx v) { v2d t; t.x = v.x; t.y = v.y; *(GLOBAL v2d*)p = t; }
__device__ inline void gsnt(cplx* p, cplx v) { v2d t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, (GLOBAL v2d*)p); }
__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kTB = 64, kTileElems = 4096, kStrip = 16;

__device__ __forceinline__ void half(const cplx (&a)[8], int h, const cplx (*__restrict__ xir)[2], const cplx (&xJ1)[2],
                                     const cplx (&xJ2)[2], cplx (&acc1)[8], cplx (&acc2)[8], cplx (&c1)[2], cplx (&c2)[2]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;
        const cplx x1 = xir[8 * j][0], x2 = xir[8 * j][1];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const cplx e = a[2 * jj + cc];
            acc1[j] = acc1[j] + e * xJ1[cc];
            acc2[j] = acc2[j] + e * xJ2[cc];
            c1[cc] = mk(c1[cc].x + e.x * x1.x + e.y * x1.y, c1[cc].y + e.x * x1.y - e.y * x1.x);
            c2[cc] = mk(c2[cc].x + e.x * x2.x + e.y * x2.y, c2[cc].y + e.x * x2.y - e.y * x2.x);
        }
    }
}

template <int F>
__global__ void __launch_bounds__(256, (F & 2) ? 3 : 2)
probe_kernel(const cplx* __restrict__ tiles, const cplx* __restrict__ vec, cplx* __restrict__ colpart, cplx* __restrict__ rowpart, int ntile) {
    constexpr bool STORE = F & 1;
    __shared__ cplx lds[(F & 2) ? 3072 : 4096];
    cplx (*xs)[kTB][2] = reinterpret_cast<cplx (*)[kTB][2]>(lds);
    cplx (*xi)[2] = reinterpret_cast<cplx (*)[2]>(lds + 2048);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3, col = 16 * w + cg;
    const cplx* __restrict__ tp = tiles + (size_t)blockIdx.x * ntile * kTileElems + rg * kTB + col;
    cplx a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = glnt(tp + (8 * (k >> 1)) * kTB + 8 * (k & 1));
    if (threadIdx.x < 2 * kTB) lds[2048 + threadIdx.x] = gl(vec + threadIdx.x);
    for (int idx = threadIdx.x; idx < ntile * 2 * kTB; idx += 256) lds[idx] = gl(vec + idx);
    __syncthreads();
    cplx acc1[8], acc2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc1[j] = mk(0.0, 0.0); acc2[j] = mk(0.0, 0.0); }
    const cplx (*__restrict__ xir)[2] = xi + rg;
    const int cslot = 2 * (col + 8 * ((rg >> 1) & 1)) + (rg & 1);
    cplx* __restrict__ cp = colpart + (size_t)blockIdx.x * ntile * 2 * kTB + cslot;
    cplx sink = mk(0.0, 0.0);
    auto tile_end = [&](cplx (&c1)[2], cplx (&c2)[2], int t) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                c1[cc] = mk(c1[cc].x + __shfl_xor(c1[cc].x, o, 64), c1[cc].y + __shfl_xor(c1[cc].y, o, 64));
                c2[cc] = mk(c2[cc].x + __shfl_xor(c2[cc].x, o, 64), c2[cc].y + __shfl_xor(c2[cc].y, o, 64));
            }
        }
        const bool v1 = rg & 1, ch1 = rg & 2;
        const double lx = v1 ? c2[0].x : c1[0].x, ly = v1 ? c2[0].y : c1[0].y;
        const double hx = v1 ? c2[1].x : c1[1].x, hy = v1 ? c2[1].y : c1[1].y;
        const cplx val = mk(ch1 ? hx : lx, ch1 ? hy : ly);
        if (F & 8) lds[2 * (t * kTB) + cslot] = val;
        else if (STORE) gs(cp + 2 * (t * kTB), val); else sink = sink + val;
    };
#pragma unroll 1
    for (int t = 0; t + 1 < ntile; ++t) {
        const cplx* __restrict__ tc = tp + (size_t)t * kTileElems;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = glnt(tc + (8 * (4 + (k >> 1))) * kTB + 8 * (k & 1));
        cplx xJ1[2], xJ2[2], c1[2], c2[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { xJ1[cc] = xs[t][col + 8 * cc][0]; xJ2[cc] = xs[t][col + 8 * cc][1]; c1[cc] = mk(0, 0); c2[cc] = mk(0, 0); }
        half(a0, 0, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) a0[k] = glnt(tc + kTileElems + (8 * (k >> 1)) * kTB + 8 * (k & 1));
        __builtin_amdgcn_sched_barrier(0);
        if (F & 4) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) { xJ1[cc] = xs[t][col + 8 * cc][0]; xJ2[cc] = xs[t][col + 8 * cc][1]; }
        }
        half(a1, 1, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        tile_end(c1, c2, t);
    }
    {
        const int t = ntile - 1;
        const cplx* __restrict__ tc = tp + (size_t)t * kTileElems;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = glnt(tc + (8 * (4 + (k >> 1))) * kTB + 8 * (k & 1));
        cplx xJ1[2], xJ2[2], c1[2], c2[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { xJ1[cc] = xs[t][col + 8 * cc][0]; xJ2[cc] = xs[t][col + 8 * cc][1]; c1[cc] = mk(0, 0); c2[cc] = mk(0, 0); }
        half(a0, 0, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        half(a1, 1, xir, xJ1, xJ2, acc1, acc2, c1, c2);
        tile_end(c1, c2, t);
    }
    __syncthreads();
    if (F & 8) for (int idx = threadIdx.x; idx < ntile * 2 * kTB; idx += 256) {
        cplx* dst = colpart + ((F & 32) ? 0 : (size_t)((F & 64) ? blockIdx.x % 512 : (F & 128) ? blockIdx.x % 2048 : blockIdx.x) * ntile * 2 * kTB) + idx;
        if (F & 16) gsnt(dst, lds[idx]); else gs(dst, lds[idx]);
    }
    __syncthreads();
    cplx* __restrict__ red = lds + w * 576;
    cplx (*rsum)[kTB][2] = reinterpret_cast<cplx (*)[kTB][2]>(lds + 2560);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int e = j * 64 + lane; red[e + (e >> 3)] = v ? acc2[j] : acc1[j]; }
        wave_lds_sync();
        cplx s = red[lane * 9];
#pragma unroll
        for (int k = 1; k < 8; ++k) s = s + red[lane * 9 + k];
        wave_lds_sync();
        rsum[w][lane][v] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2 * kTB) {
        const int row = threadIdx.x >> 1, v = threadIdx.x & 1;
        gs(rowpart + (size_t)blockIdx.x * 2 * kTB + threadIdx.x, ((rsum[0][row][v] + rsum[1][row][v]) + rsum[2][row][v]) + rsum[3][row][v]);
    }
    if (sink.x == 1.2345e300) gs(rowpart, sink);
}

__global__ void fill_kernel(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
        p[i] = (double)(h & 0xFFFFF) * (1.0 / 1048576.0) - 0.5;
    }
}

static int g_reps = 20;
template <int F>
static void run(const cplx* buf, size_t bytes, const cplx* vec, cplx* colpart, cplx* rowpart, int ntile, const char* what) {
    const int nwg = (int)(bytes / ((size_t)ntile * 65536));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_kernel<F>, dim3(nwg), dim3(256), 0, 0, buf, vec, colpart, rowpart, ntile);
    hipEventRecord(e0, 0);
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(probe_kernel<F>, dim3(nwg), dim3(256), 0, 0, buf, vec, colpart, rowpart, ntile);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("flags %2d  tiles/strip %2d  %-52s %7.1f GB/s\n", F, ntile, what, (double)nwg * ntile * 65536.0 * g_reps / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    if (argc > 1) g_reps = atoi(argv[1]);
    const size_t bytes = (size_t)3 << 30;
    cplx *buf, *vec, *colpart, *rowpart;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&vec, 1 << 20) != hipSuccess ||
        hipMalloc(&colpart, (bytes / 65536) * kTB * 2 * 16) != hipSuccess || hipMalloc(&rowpart, (bytes / 65536) * kTB * 2 * 16 + 64) != hipSuccess) {
        printf("alloc failed\n"); return 1;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (double*)buf, bytes / 8);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (double*)vec, (size_t)(1 << 20) / 8);
    hipDeviceSynchronize();
    run<0>(buf, bytes, vec, colpart, rowpart, 16, "no column stores");
    run<1>(buf, bytes, vec, colpart, rowpart, 16, "product form");
    run<1>(buf, bytes, vec, colpart, rowpart, 8, "product form");
    run<1>(buf, bytes, vec, colpart, rowpart, 4, "product form");
    run<8>(buf, bytes, vec, colpart, rowpart, 16, "column partials through LDS, one burst per strip");
    run<8>(buf, bytes, vec, colpart, rowpart, 8, "column partials through LDS, one burst per strip");
    run<8>(buf, bytes, vec, colpart, rowpart, 4, "column partials through LDS, one burst per strip");
    run<8 + 16>(buf, bytes, vec, colpart, rowpart, 16, "... as non-temporal stores");
    run<8 + 32>(buf, bytes, vec, colpart, rowpart, 16, "... into one L2-resident region");
    run<8 + 64>(buf, bytes, vec, colpart, rowpart, 16, "... into a 16 MB ring (512 regions)");
    run<8 + 128>(buf, bytes, vec, colpart, rowpart, 16, "... into a 64 MB ring (2048 regions)");
    run<3>(buf, bytes, vec, colpart, rowpart, 16, "three workgroups per CU");
    return 0;
}
