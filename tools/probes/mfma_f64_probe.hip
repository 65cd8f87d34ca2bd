// Stand-alone probe for v_mfma_f64_16x16x4_f64 on the GPU at hand (next to the wide-block kernels of
// scintools_amd/csrc/blockw_kernels.hpp, which assume the operand layout printed below):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// 1. layout: D = C + A B with A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, result register r of
//    lane l = row (l >> 4) + 4 r, column l & 15 -- checked element by element against the host;
// 2. rate: independent and dependent chains of MFMAs per wave, cycles per instruction from the clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* a, const double* b, const double* c, double* d) {
    const int lane = threadIdx.x;
    v4d acc = {c[4 * lane], c[4 * lane + 1], c[4 * lane + 2], c[4 * lane + 3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[4 * lane + r] = acc[r];
}

template <int CHAINS>
__global__ void rate_kernel(double* out, int iters, long long* cycles) {
    v4d acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    const long long t1 = clock64();
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    double ha[64], hb[64], hc[256], hd[256];
    srand(1);
    for (int i = 0; i < 64; ++i) { ha[i] = rand() / (double)RAND_MAX - 0.5; hb[i] = rand() / (double)RAND_MAX - 0.5; }
    for (int i = 0; i < 256; ++i) hc[i] = rand() / (double)RAND_MAX - 0.5;
    double *a, *b, *c, *d;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&c, sizeof(hc)); hipMalloc(&d, sizeof(hd));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(c, hc, sizeof(hc), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, c, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int i = (lane >> 4) + 4 * r, j = lane & 15;
            double ref = hc[4 * lane + r];
            for (int k = 0; k < 4; ++k) ref = fma(ha[16 * k + i], hb[16 * k + j], ref);
            if (fabs(ref - hd[4 * lane + r]) > 1e-13) ++bad;
        }
    printf("layout (A[i][k] lane 16k+i, B[k][j] lane 16k+j, D reg r of lane l = row (l>>4)+4r, col l&15): %s (%d of 256 differ)\n",
           bad ? "MISMATCH" : "ok", bad);
    if (bad) {   // try the f32-style result map to tell which one the hardware uses
        int bad2 = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * (lane >> 4) + r, j = lane & 15;
                double ref = 0.0;
                for (int k = 0; k < 4; ++k) ref = fma(ha[16 * k + i], hb[16 * k + j], ref);
                // C in the same (wrong-for-us) map cannot be separated here: compare without it
                if (fabs(ref + hc[4 * lane + r] - hd[4 * lane + r]) > 1e-13) ++bad2;
            }
        printf("  with the f32 map row 4 (l>>4) + r instead: %d of 256 differ\n", bad2);
    }
    double* out; long long* cyc; long long hcyc = 0;
    hipMalloc(&out, sizeof(double) * 256 * 1024); hipMalloc(&cyc, sizeof(long long));
    const int iters = 20000;
    hipLaunchKernelGGL(rate_kernel<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc); hipMemcpy(&hcyc, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave, 1 dependent chain : %.1f clocks per MFMA\n", (double)hcyc / iters);
    hipLaunchKernelGGL(rate_kernel<4>, dim3(1), dim3(64), 0, 0, out, iters, cyc); hipMemcpy(&hcyc, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave, 4 independent     : %.1f clocks per MFMA\n", (double)hcyc / (4.0 * iters));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<4>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);      // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<4>, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 1024.0 * 4 /*waves*/ * 4.0 * iters * 2048.0;
    printf("whole GPU, 4096 waves x 4 chains: %.1f TFLOP/s fp64 (matrix)\n", flops / (ms * 1e-3) / 1e12);
    return bad ? 1 : 0;
}
