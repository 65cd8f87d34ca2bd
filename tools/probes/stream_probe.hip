// Stand-alone probe: what does the LOAD STRUCTURE of the packed mat-vec kernels reach on this GPU,
// with the arithmetic taken out?  Every workgroup streams `tiles` contiguous 64-KiB tiles exactly as
// pk2_matvec_kernel does (4 waves, each wave 16 rows of 1 KiB per tile in two halves of H loads, the
// next half requested before the current one is consumed), sums what it read and writes one value.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
// Variables: loads in flight per wave (2 H), non-temporal or plain loads, workgroups per CU (set by a
// dummy dynamic LDS size), tiles per workgroup.  Prints GB/s of each configuration over a 3 GiB buffer
// (beyond the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v2d __attribute__((ext_vector_type(2)));
#define GLOBAL __attribute__((address_space(1)))

template <bool NT>
__device__ inline v2d ld(const v2d* p) {
    if (NT) return __builtin_nontemporal_load((const GLOBAL v2d*)p);
    return *(const GLOBAL v2d*)p;
}

// H rows of 64 x 16 B per half, HALVES halves per tile-iteration: a wave covers 16 rows of a 64-KiB tile
// (H = 8), or 32 rows of two consecutive tiles' worth (H = 16: twice the bytes in flight)
template <int H, bool NT>
__global__ void __launch_bounds__(256) stream_kernel(const v2d* __restrict__ buf, int tiles, double* out) {
    extern __shared__ char dummy[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t wg_elems = (size_t)tiles * 4096;
    const v2d* __restrict__ p = buf + (size_t)blockIdx.x * wg_elems + (size_t)(2 * H * w) * 64 + lane;
    const int iters = tiles * 16 / (2 * H);            // wave-iterations: each covers 2 H rows x 4 waves
    v2d a0[H], a1[H];
#pragma unroll
    for (int r = 0; r < H; ++r) a0[r] = ld<NT>(p + r * 64);
    v2d s = {0.0, 0.0};
#pragma unroll 1
    for (int t = 0; t < iters; ++t) {
        const v2d* __restrict__ tc = p + (size_t)t * (8 * H * 64);
#pragma unroll
        for (int r = 0; r < H; ++r) a1[r] = ld<NT>(tc + (H + r) * 64);
        __builtin_amdgcn_sched_barrier(0);       // keep the structure: request, then consume the older half
#pragma unroll
        for (int r = 0; r < H; ++r) s += a0[r];
        __builtin_amdgcn_sched_barrier(0);
        const v2d* __restrict__ nx = t + 1 < iters ? tc + 8 * H * 64 : tc;
#pragma unroll
        for (int r = 0; r < H; ++r) a0[r] = ld<NT>(nx + r * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < H; ++r) s += a1[r];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (s.x == 1.2345e300) out[blockIdx.x] = s.y + dummy[0];
}

template <int H, bool NT>
static void run(const v2d* buf, size_t bytes, int tiles, int lds_kib, double* out) {
    const size_t wg_bytes = (size_t)tiles * 65536;
    const int nwg = (int)(bytes / wg_bytes);
    auto k = stream_kernel<H, NT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds_kib * 1024, 0, buf, tiles, out);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds_kib * 1024, 0, buf, tiles, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("loads/wave %2d  %s  tiles/wg %3d  lds %3d KiB (%d wg/CU)  %7.1f GB/s\n", 2 * H, NT ? "nt   " : "plain", tiles, lds_kib,
           lds_kib ? 160 / lds_kib : 8, (double)nwg * wg_bytes * reps / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)3 << 30;
    v2d* buf; double* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    for (int tiles : {16, 4, 64}) {
        for (int lds : {64, 40, 32, 16}) {
            run<8, true>(buf, bytes, tiles, lds, out);
            run<8, false>(buf, bytes, tiles, lds, out);
        }
        run<16, true>(buf, bytes, tiles, 64, out);
        run<16, true>(buf, bytes, tiles, 32, out);
        run<4, true>(buf, bytes, tiles, 32, out);
        run<4, true>(buf, bytes, tiles, 16, out);
    }
    return 0;
}
