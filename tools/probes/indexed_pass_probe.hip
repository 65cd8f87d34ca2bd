// Stand-alone probe (round 6): an INDEX-COMPRESSED pass of the eigenvalue sweep.  The library streams the gathered
// theta-theta matrix of every curvature from HBM on every Lanczos pass (16 bytes per strict-upper element).  Here a pass streams
// a 4-byte offset per element instead and fetches the payload from the conjugate spectrum itself (4096^2 complex128 = 268 MB, the
// touched half 134 MB: resident in the 256-MiB Infinity Cache and shared by all resident curvatures), times a weight from a
// |j - i| table -- the same two-vector arithmetic, tile loop and column-partial handling as pk2_matvec_kernel (pk3_probe.hip, NV = 2).
// Offsets are the gather's own (ththmod.py:94-104) for a 4096-edge uniform theta grid at eta = 0.25 / 1 / 4 eta_true of the headline
// geometry; tiles are the 64 x 64 tiles of block rows I = 0..63, J = I + t (wrapped): the locality of a real pass.
// Reported: elements x 16 bytes / time ("equivalent stream rate") beside the streamed loop's rate in the same process; the
// experiment is kept only above 1.15x.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/indexed_pass_probe.hip -o /tmp/indexed_pass_probe && /tmp/indexed_pass_probe [reps]
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

template <int NV, int WG, bool INDEXED>
__global__ void __launch_bounds__(256, WG)
probe_kernel(const cplx* __restrict__ tiles, const cplx* __restrict__ vec, cplx* __restrict__ colpart, cplx* __restrict__ rowpart, int ntile,
             const int* __restrict__ offs, const cplx* __restrict__ cs, const double* __restrict__ wtab) {
    extern __shared__ cplx lds[];                       // xs [ntile][64][NV] | xi [64][NV] | reduction scratch
    cplx* xs = lds;
    cplx* xi = lds + ntile * kTB * NV;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cg = lane & 7, rg = lane >> 3, col = 16 * w + cg;
    const cplx* __restrict__ tp = tiles + (size_t)blockIdx.x * ntile * kTileElems + rg * kTB + col;
    const int* __restrict__ op = offs + (size_t)blockIdx.x * ntile * kTileElems + rg * kTB + col;
    // element (row rg + 8 q, column col + 8 cc) of tile t of this strip: streamed, or offset -> conjugate spectrum x weight[j - i]
    const int I = (int)blockIdx.x % 64;
    auto fetch = [&](int t, int q, int cc) {
        const int e = (8 * q) * kTB + 8 * cc;
        if (!INDEXED) return glnt(tp + (size_t)t * kTileElems + e);
        const int o = __builtin_nontemporal_load(op + (size_t)t * kTileElems + e);
        const int dj = 64 * (((I + t) & 63) - I) + (col + 8 * cc) - (rg + 8 * q);
        const double wgt = wtab[dj < 0 ? -dj : dj];
        const cplx v = gl(cs + o);
        return mk(v.x * wgt, v.y * wgt);
    };
    cplx a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = fetch(0, k >> 1, k & 1);
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
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = fetch(t, 4 + (k >> 1), k & 1);
        cplx xJ[NV][2], c[NV][2];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) { xJ[v][cc] = xs[NV * (t * kTB + col + 8 * cc) + v]; c[v][cc] = mk(0, 0); }
        half<NV>(a0, 0, xir, xJ, acc, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) a0[k] = fetch(t + 1, k >> 1, k & 1);
        __builtin_amdgcn_sched_barrier(0);
        half<NV>(a1, 1, xir, xJ, acc, c);
        tile_end(c, t);
    }
    {
        const int t = ntile - 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) a1[k] = fetch(t, 4 + (k >> 1), k & 1);
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

// offsets of the gather (ththmod.py:94-104) for the probe's tiles: strip b = (copy, block row I), tile t -> J = (I + t) mod 64
__global__ void offsets_kernel(int* offs, int nstrips, int ntile, double eta, double th0, double dth, double tau0, double dtau, double fd0, double dfd, int n) {
    const size_t total = (size_t)nstrips * ntile * kTileElems;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int e = (int)(g % kTileElems), t = (int)((g / kTileElems) % ntile), b = (int)(g / ((size_t)kTileElems * ntile));
        const int I = b % 64, J = (I + t) & 63, r = e / kTB, c = e % kTB;
        const double th2 = th0 + dth * (64 * I + r), th1 = th0 + dth * (64 * J + c);     // row = theta_2, column = theta_1
        long ti = (long)floor((eta * (th1 * th1 - th2 * th2) - tau0 + dtau / 2) / dtau);
        long fi = (long)floor(((th1 - th2) - fd0 + dfd / 2) / dfd);
        ti = ti < 0 ? 0 : (ti >= n ? n - 1 : ti);
        fi = fi < 0 ? 0 : (fi >= n ? n - 1 : fi);
        offs[g] = (int)(ti * n + fi);
    }
}
__global__ void wtab_kernel(double* w, int n, double two_eta_dth) { for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) w[k] = sqrt(two_eta_dth * k); }

static int g_reps = 20;
template <bool INDEXED>
static double run(const cplx* buf, const cplx* vec, cplx* colpart, cplx* rowpart, int nstrips, int ntile, const int* offs, const cplx* cs, const double* wtab, const char* what) {
    const size_t lds = sizeof(cplx) * (size_t)std::max((ntile + 1) * kTB * 2, 4 * 576 + 4 * kTB * 2);
    auto k = probe_kernel<2, 2, INDEXED>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nstrips), dim3(256), lds, 0, buf, vec, colpart, rowpart, ntile, offs, cs, wtab);
    hipEventRecord(e0, 0);
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(k, dim3(nstrips), dim3(256), lds, 0, buf, vec, colpart, rowpart, ntile, offs, cs, wtab);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)nstrips * ntile * 65536.0 * g_reps / (ms * 1e-3) / 1e9;
    printf("%-64s %7.1f GB/s (16 B per element)\n", what, gbs);
    return gbs;
}

int main(int argc, char** argv) {
    if (argc > 1) g_reps = atoi(argv[1]);
    const int n = 4096, ntile = 16, copies = 24, nstrips = 64 * copies;          // 24 "curvatures" resident: 1536 workgroups, 1.6 GB of tiles
    const size_t elems = (size_t)nstrips * ntile * kTileElems;
    cplx *buf, *vec, *colpart, *rowpart, *cs; int* offs; double* wtab;
    if (hipMalloc(&buf, elems * 16) != hipSuccess || hipMalloc(&vec, 1 << 20) != hipSuccess || hipMalloc(&cs, (size_t)n * n * 16) != hipSuccess ||
        hipMalloc(&offs, elems * 4) != hipSuccess || hipMalloc(&wtab, n * 8) != hipSuccess ||
        hipMalloc(&colpart, (size_t)nstrips * ntile * kTB * 2 * 16) != hipSuccess || hipMalloc(&rowpart, (size_t)nstrips * kTB * 2 * 16 + 64) != hipSuccess) {
        printf("alloc failed\n"); return 1;
    }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (double*)buf, elems * 2);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (double*)cs, (size_t)n * n * 2);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (double*)vec, (size_t)(1 << 20) / 8);
    // the headline geometry: fd.max 16.66 mHz, tau.max 5.85 us, eta_true 0.0202 s^3, theta in +-8.33 mHz
    const double fdmax = 16.66, taumax = 5.85, eta_true = 0.0202, dfd = 2 * fdmax / n, dtau = 2 * taumax / n, th0 = -fdmax / 2, dth = fdmax / n;
    for (int rep = 0; rep < 2; ++rep) {
        const double base = run<false>(buf, vec, colpart, rowpart, nstrips, ntile, offs, cs, wtab, "streamed tiles (the library's pass), two vectors");
        for (double f : {0.25, 1.0, 4.0}) {
            hipLaunchKernelGGL(offsets_kernel, dim3(4096), dim3(256), 0, 0, offs, nstrips, ntile, f * eta_true, th0, dth, -taumax, dtau, -fdmax, dfd, n);
            hipLaunchKernelGGL(wtab_kernel, dim3(16), dim3(256), 0, 0, wtab, n, 2 * f * eta_true * dth);
            hipDeviceSynchronize();
            char what[128];
            snprintf(what, sizeof(what), "indexed pass, eta = %.2f eta_true (4 B offsets + CS gather + weight)", f);
            const double g = run<true>(buf, vec, colpart, rowpart, nstrips, ntile, offs, cs, wtab, what);
            printf("   ratio to the streamed pass: %.3f   (kept only above 1.15)\n", g / base);
        }
    }
    return 0;
}
