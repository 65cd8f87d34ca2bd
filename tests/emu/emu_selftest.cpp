// TEST INFRASTRUCTURE ONLY: small kernels that pin the interpreter's own semantics (cross-lane
// operations, barriers, the f64 matrix-core model) against NumPy -- tests/test_emu_cpu.py.
#include <hip/hip_runtime.h>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void selftest_mfma_kernel(const double* a, const double* b, const double* c, double* d) {
    const int lane = threadIdx.x;
    v4d acc = {c[4 * lane], c[4 * lane + 1], c[4 * lane + 2], c[4 * lane + 3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[lane], b[lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[4 * lane + r] = acc[r];
}

// out[0..63]: xor-butterfly sum; [64..127]: shfl from lane (l * 7) % 64; [128..191]: readlane 5 of the
// wave's id-dependent value; [192..255]: ballot bit count of (l % 3 == 0) among lanes < limit (others
// skip the ballot: divergent participation); [256..319]: block-wide LDS exchange across 4 waves
__global__ void selftest_wave_kernel(const double* in, double* out, int limit) {
    __shared__ double lds[256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double v = in[t];
    double s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double g = __shfl(v, (lane * 7) % 64, 64);
    const int rl = __builtin_amdgcn_readlane((int)(v * 1000.0), 5);
    int cnt = -1;
    if (lane < limit) cnt = __popcll(__ballot(lane % 3 == 0));
    lds[t] = v;
    __syncthreads();
    const double x = lds[(t + 64) % 256];          // the next wave's value
    if (w == 0) {
        out[lane] = s; out[64 + lane] = g; out[128 + lane] = (double)rl; out[192 + lane] = (double)cnt;
        out[256 + lane] = x;
    }
}

extern "C" int emu_selftest_mfma(const double* a, const double* b, const double* c, double* d) {
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, nullptr, a, b, c, d);
    return 0;
}
extern "C" int emu_selftest_wave(const double* in, double* out, int limit) {
    hipLaunchKernelGGL(selftest_wave_kernel, dim3(1), dim3(256), 0, nullptr, in, out, limit);
    return 0;
}
