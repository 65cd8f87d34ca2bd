// Stand-alone probe: issue cost of the fp64 vector instructions the FFT kernels are made of, per wave64 instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fp64_rate_probe.hip -o /tmp/fp64_rate_probe && /tmp/fp64_rate_probe
// Each thread runs CH independent chains of N dependent operations; 1, 2 or 4 waves per SIMD (256, 512, 1024 threads per
// workgroup, one workgroup per CU).  Prints shader cycles (s_memtime) per wave instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int OP, int CH>
__global__ void k(double* out, int n, double a, double b) {
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = a + c + threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
                if (OP == 1) x[c] = x[c] * a;
                if (OP == 2) x[c] = x[c] + b;
                if (OP == 3) x[c] = (double)((float)x[c] * (float)a + (float)b);   // f32 for scale (with conversions)
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)(t1 - t0) * 1e-300;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

template <int OP, int CH>
void run(const char* name, double* d) {
    for (int threads : {256, 512, 1024}) {
        const int n = 2000;
        hipLaunchKernelGGL((k<OP, CH>), dim3(256), dim3(threads), 0, 0, d, n, 1.0000001, 1e-9);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, CH>), dim3(256), dim3(threads), 0, 0, d, n, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
        const double insts_per_wave = (double)n * 8 * CH, waves_per_simd = threads / 256.0;
        printf("%-10s chains=%d waves/SIMD=%.0f: %.2f counter ticks per wave instruction per SIMD; %.3f ms -> %.2f ns per instruction per SIMD\n", name, CH,
               waves_per_simd, cyc / (insts_per_wave * waves_per_simd), ms, ms * 1e6 / (insts_per_wave * waves_per_simd));
    }
}
int main() {
    double* d; hipMalloc(&d, 256 * 1024 * 8);
    run<0, 8>("fma_f64", d); run<0, 1>("fma_f64", d);
    run<1, 8>("mul_f64", d);
    run<2, 8>("add_f64", d);
    return 0;
}
