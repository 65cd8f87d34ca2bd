// Stand-alone probe (round 6): the mat-vec loop of pk2_matvec_kernel (scintools_amd/csrc/eigen_packed.hip) with a block
// of NV = 2 (the library's) or 3 vectors, arithmetic included, on synthetic strips as in pk2e_probe.hip (every workgroup: NT contiguous
// 64-KiB tiles of a 3 GiB buffer; column partials through LDS, one burst per strip -- the adopted form).  A three-vector block Lanczos
// needs 0.87x (analytic arc) / 0.91x (reference Simulation screen) of the two-vector block's matrix bytes
// (profiles/r05_block_width_model.txt): it pays only if this loop streams at more than that fraction of the two-vector rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk3_probe.hip -o /tmp/pk3_probe && /tmp/pk3_probe [reps]
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
__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kTB = 64, kTileElems = 4096;

// rows 8 j + rg (j = 4 h .. 4 h + 3) of the tile against the wave's two columns: row sums (acc) and column partials (c), NV vectors
template <int NV>
__device__ __forceinline__ void half(const cplx (&a)[8], int h, const cplx* __restrict__ xir, const cplx (&xJ)[NV][2],
                                     cplx (&acc)[NV][8], cplx (&c)[NV][2]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * h + jj;
        cplx x[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] = xir[(8 * j) * NV + v];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const cplx e = a[2 * jj + cc];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                acc[v][j] = acc[v][j] + e * xJ[v][cc];
                c[v][cc] = mk(c[v][cc].x + e.x * x[v].x + e.y * x[v].y, c[v][cc].y + e.x * x[v].y - e.y * x[v].x);
            }
        }
    }
}

template <int NV, int WG>
__global__ void __launch_bounds__(256, WG)
probe_kernel(const cplx* __restrict__ tiles, const cplx* __restrict__ vec, cplx* __restrict__ colpart, cplx* __restrict__ rowpart, int ntile) {
    extern __shared__ cplx lds[];                       // xs [ntile][64][NV] | xi [64][NV] | reduction scratch
    cplx* xs = lds;
    cplx* xi = lds + ntile * kTB * NV;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3, col = 16 * w + cg;
    const cplx* __restrict__ tp = tiles + (size_t)blockIdx.x * ntile * kTileElems + rg * kTB + col;
    cplx a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = glnt(tp + (8 * (k >> 1)) * kTB + 8 * (k & 1));
    for (int idx = threadIdx.x; idx < (ntile + 1) * kTB * NV; idx += 256) lds[idx] = gl(vec + idx);
    __syncthreads();
    cplx acc[NV][8];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[v][j] = mk(0.0, 0.0);
    const cplx* __restrict__ xir = xi + rg * NV;
    // which (vector, column half) this row group writes back: NV x 2 values, eight row groups
    const int vsel = rg % NV, hsel = (rg / NV) & 1;
    const bool writes = rg < 2 * NV;
    const int cslot = NV * (col + 8 * hsel) + vsel;
    auto tile_end = [&](cplx (&c)[NV][2], int t) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
                    c[v][cc] = mk(c[v][cc].x + __shfl_xor(c[v][cc].x, o, 64), c[v][cc].y + __shfl_xor(c[v][cc].y, o, 64));
        cplx val = c[0][0];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                if (vsel == v && hsel == cc) val = c[v][cc];
        if (writes) xs[NV * (t * kTB) + cslot] = val;     // into the consumed x_J slots
    };
#pragma unroll 1
    for (int t = 0; t + 1 < ntile; ++t) {
        const cplx* __restrict__ tc = tp + (size_t)t * kTileElems;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = glnt(tc + (8 * (4 + (k >> 1))) * kTB + 8 * (k & 1));
        cplx xJ[NV][2], c[NV][2];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) { xJ[v][cc] = xs[NV * (t * kTB + col + 8 * cc) + v]; c[v][cc] = mk(0, 0); }
        half<NV>(a0, 0, xir, xJ, acc, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) a0[k] = glnt(tc + kTileElems + (8 * (k >> 1)) * kTB + 8 * (k & 1));
        __builtin_amdgcn_sched_barrier(0);
        half<NV>(a1, 1, xir, xJ, acc, c);
        tile_end(c, t);
    }
    {
        const int t = ntile - 1;
        const cplx* __restrict__ tc = tp + (size_t)t * kTileElems;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = glnt(tc + (8 * (4 + (k >> 1))) * kTB + 8 * (k & 1));
        cplx xJ[NV][2], c[NV][2];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) { xJ[v][cc] = xs[NV * (t * kTB + col + 8 * cc) + v]; c[v][cc] = mk(0, 0); }
        half<NV>(a0, 0, xir, xJ, acc, c);
        half<NV>(a1, 1, xir, xJ, acc, c);
        tile_end(c, t);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < ntile * NV * kTB; idx += 256) gs(colpart + (size_t)blockIdx.x * ntile * NV * kTB + idx, xs[idx]);
    __syncthreads();
    cplx* __restrict__ red = lds + w * 576;                               // (the x_J slots are consumed)
    cplx* rsum = lds + 4 * 576;                                            // [4][64][NV]
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int e = j * 64 + lane; red[e + (e >> 3)] = acc[v][j]; }
        wave_lds_sync();
        cplx s = red[lane * 9];
#pragma unroll
        for (int k = 1; k < 8; ++k) s = s + red[lane * 9 + k];
        wave_lds_sync();
        rsum[(w * kTB + lane) * NV + v] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV * kTB) {
        const int row = threadIdx.x / NV, v = threadIdx.x % NV;
        gs(rowpart + (size_t)blockIdx.x * NV * kTB + threadIdx.x,
           ((rsum[(0 * kTB + row) * NV + v] + rsum[(1 * kTB + row) * NV + v]) + rsum[(2 * kTB + row) * NV + v]) + rsum[(3 * kTB + row) * NV + v]);
    }
}

__global__ void fill_kernel(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
        p[i] = (double)(h & 0xFFFFF) * (1.0 / 1048576.0) - 0.5;
    }
}

static int g_reps = 20;
template <int NV, int WG>
static double run(const cplx* buf, size_t bytes, const cplx* vec, cplx* colpart, cplx* rowpart, int ntile) {
    const int nwg = (int)(bytes / ((size_t)ntile * 65536));
    const size_t lds = sizeof(cplx) * (size_t)std::max((ntile + 1) * kTB * NV, 4 * 576 + 4 * kTB * NV);
    auto k = probe_kernel<NV, WG>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, buf, vec, colpart, rowpart, ntile);
    hipEventRecord(e0, 0);
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, buf, vec, colpart, rowpart, ntile);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)nwg * ntile * 65536.0 * g_reps / (ms * 1e-3) / 1e9;
    printf("vectors %d  workgroups/CU (launch bound) %d  tiles/strip %2d  LDS %5.1f KiB  %7.1f GB/s\n", NV, WG, ntile, lds / 1024.0, gbs);
    return gbs;
}

int main(int argc, char** argv) {
    if (argc > 1) g_reps = atoi(argv[1]);
    const size_t bytes = (size_t)3 << 30;
    cplx *buf, *vec, *colpart, *rowpart;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&vec, 1 << 20) != hipSuccess ||
        hipMalloc(&colpart, (bytes / 65536) * kTB * 3 * 16) != hipSuccess || hipMalloc(&rowpart, (bytes / 65536) * kTB * 3 * 16 + 64) != hipSuccess) {
        printf("alloc failed\n"); return 1;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (double*)buf, bytes / 8);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (double*)vec, (size_t)(1 << 20) / 8);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep)
        for (int nt : {16, 8}) {
            const double g2 = run<2, 2>(buf, bytes, vec, colpart, rowpart, nt);
            const double g3 = run<3, 2>(buf, bytes, vec, colpart, rowpart, nt);
            const double g31 = run<3, 1>(buf, bytes, vec, colpart, rowpart, nt);
            printf("   three vectors / two vectors: %.3f (two workgroups per CU), %.3f (one, 512 registers)   -- break-even 0.87 (arc) / 0.91 (screen)\n", g3 / g2, g31 / g2);
        }
    return 0;
}
