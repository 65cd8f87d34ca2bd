// TEST INFRASTRUCTURE ONLY -- not part of the product, never loaded by scintools_amd.
//
// A stand-in for <hip/hip_runtime.h> that lets the kernel SOURCES under scintools_amd/csrc be
// compiled for the x86 host and interpreted one workgroup at a time (tests/emu/emu_runtime.cpp):
// every GPU thread is a fiber, wavefronts are 64 consecutive fibers, __syncthreads and the
// cross-lane operations (__shfl*, __ballot, v_readlane / v_readfirstlane) are scheduling points
// that the interpreter resolves when every lane of the wave (or block) has arrived.  "Device"
// memory is host memory; streams execute synchronously in enqueue order (a legal schedule).
// It exists so that the control flow and arithmetic of the kernels can be checked in a container
// without a GPU (tests/test_emu_cpu.py); it says nothing about performance, and the GPU parity
// tests (-m gpu) remain the parity evidence.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <tuple>
#include <utility>

// ---- function / variable qualifiers ---------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static          // blocks run one after the other: block-shared == static
#define __launch_bounds__(...)
#define __constant__ static

// ---- geometry ---------------------------------------------------------------------------------
struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;
constexpr int warpSize = 64;

// ---- runtime types ------------------------------------------------------------------------------
typedef enum {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotReady = 600,
    hipErrorUnknown = 999
} hipError_t;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
typedef enum {
    hipMemcpyHostToHost = 0,
    hipMemcpyHostToDevice = 1,
    hipMemcpyDeviceToHost = 2,
    hipMemcpyDeviceToDevice = 3,
    hipMemcpyDefault = 4
} hipMemcpyKind;
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void** p, size_t bytes);
template <typename T> inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags = 0);
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) {
    return hipHostMalloc((void**)p, bytes, flags);
}
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* func, hipFuncAttribute attr, int value);

// ---- the interpreter ----------------------------------------------------------------------------
namespace emu {
enum WaveOp { kShfl = 1, kShflXor, kShflDown, kShflUp, kBallot, kReadFirst };
uint64_t wave_op(int op, uint64_t value, int arg, int width);
// every active lane deposits two values; returns the wave's snapshot [64][2] (valid until the wave's
// next gather) -- the building block of the matrix-core instructions
const uint64_t (*wave_gather2(uint64_t v0, uint64_t v1))[2];
int lane_id();
void block_barrier();
void run_grid(dim3 grid, dim3 block, size_t shmem, void (*thunk)(void*), void* ctx);

template <typename T> inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "cross-lane value wider than 64 bits");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <typename T> inline T from_bits(uint64_t b) {
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}
template <typename... P> struct Call {
    void (*k)(P...);
    std::tuple<P...> args;
    static void thunk(void* self) {
        Call* c = (Call*)self;
        std::apply(c->k, c->args);
    }
};
template <typename... P, typename... A>
inline void launch_kernel(void (*k)(P...), dim3 grid, dim3 block, size_t shmem, A&&... a) {
    Call<P...> c{k, std::tuple<P...>(static_cast<P>(std::forward<A>(a))...)};
    run_grid(grid, block, shmem, &Call<P...>::thunk, &c);
}
}  // namespace emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::emu::launch_kernel(kernel, dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)

inline void __syncthreads() { ::emu::block_barrier(); }

template <typename T> inline T __shfl(T v, int src, int width = 64) {
    return ::emu::from_bits<T>(::emu::wave_op(::emu::kShfl, ::emu::to_bits(v), src, width));
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    return ::emu::from_bits<T>(::emu::wave_op(::emu::kShflXor, ::emu::to_bits(v), mask, width));
}
template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
    return ::emu::from_bits<T>(::emu::wave_op(::emu::kShflDown, ::emu::to_bits(v), (int)delta, width));
}
template <typename T> inline T __shfl_up(T v, unsigned delta, int width = 64) {
    return ::emu::from_bits<T>(::emu::wave_op(::emu::kShflUp, ::emu::to_bits(v), (int)delta, width));
}
inline unsigned long long __ballot(int pred) { return ::emu::wave_op(::emu::kBallot, pred != 0, 0, 64); }
inline int emu_readlane(int v, int lane) {
    return ::emu::from_bits<int>(::emu::wave_op(::emu::kShfl, ::emu::to_bits(v), lane, 64));
}
inline int emu_readfirstlane(int v) {
    return ::emu::from_bits<int>(::emu::wave_op(::emu::kReadFirst, ::emu::to_bits(v), 0, 64));
}
// v_mfma_f64_16x16x4_f64: D = C + A(16x4) B(4x16).  Operand layout (cdna_hip_programming.md, f64
// MFMA): A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, C/D register r of lane l holds
// row (l >> 4) + 4 r, column l & 15.
typedef double emu_v4d __attribute__((ext_vector_type(4)));
inline emu_v4d emu_mfma_f64_16x16x4(double a, double b, emu_v4d c) {
    const uint64_t (*snap)[2] = ::emu::wave_gather2(::emu::to_bits(a), ::emu::to_bits(b));
    const int lane = ::emu::lane_id(), j = lane & 15;
    emu_v4d d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = __builtin_fma(::emu::from_bits<double>(snap[16 * k + i][0]), ::emu::from_bits<double>(snap[16 * k + j][1]), acc);
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, cbsz, abid, blgp) emu_mfma_f64_16x16x4((a), (b), (c))
// v_mov_b32_dpp row_newbcast:K (dpp_ctrl 0x150 + K): lane K of each 16-lane row to the whole row
inline int emu_update_dpp(int v, int ctrl) {
    const int lane = ::emu::lane_id();
    return ::emu::from_bits<int>(::emu::wave_op(::emu::kShfl, ::emu::to_bits(v), (lane & ~15) | (ctrl & 15), 64));
}
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rmask, bmask, bctl) emu_update_dpp((v), (ctrl))
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)     // instruction scheduling only
#define __builtin_amdgcn_rcp(x) (1.0 / (x))                 // v_rcp_f64: an approximation on the GPU, refined by its callers
#define __builtin_amdgcn_readlane(v, lane) emu_readlane((v), (lane))
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)

// buffer-resource accesses: base + per-lane byte offset + scalar byte offset.  The hardware's range check covers the per-lane
// offset only (out of range reads 0) and not the scalar one, so the kernels must not rely on it: here an access beyond the
// resource aborts the test.
struct __amdgpu_buffer_rsrc_t { char* base; uint32_t bytes; };
inline __amdgpu_buffer_rsrc_t emu_make_buffer_rsrc(void* p, uint32_t bytes) { return __amdgpu_buffer_rsrc_t{(char*)p, bytes}; }
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) emu_make_buffer_rsrc((p), (uint32_t)(bytes))
typedef unsigned int emu_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int emu_v2u __attribute__((ext_vector_type(2)));
template <typename V> inline V emu_buffer_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    V v = {};
    const uint64_t o = (uint64_t)(uint32_t)voff + (uint64_t)(uint32_t)soff;
    if (o + sizeof(V) > r.bytes) { fprintf(stderr, "emu: buffer load beyond the resource (%llu + %zu > %u)\n", (unsigned long long)o, sizeof(V), r.bytes); abort(); }
    memcpy(&v, r.base + o, sizeof(V));
    return v;
}
template <typename V> inline void emu_buffer_store(V v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const uint64_t o = (uint64_t)(uint32_t)voff + (uint64_t)(uint32_t)soff;
    if (o + sizeof(V) > r.bytes) { fprintf(stderr, "emu: buffer store beyond the resource (%llu + %zu > %u)\n", (unsigned long long)o, sizeof(V), r.bytes); abort(); }
    memcpy(r.base + o, &v, sizeof(V));
}
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) emu_buffer_load<emu_v4u>((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) emu_buffer_load<emu_v2u>((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, aux) emu_buffer_store<emu_v4u>((v), (r), (voff), (soff))

// ---- device library bits the kernels use --------------------------------------------------------
inline int __double2hiint(double v) { return (int)(::emu::to_bits(v) >> 32); }
inline int __double2loint(double v) { return (int)(::emu::to_bits(v) & 0xffffffffu); }
inline double __hiloint2double(int hi, int lo) {
    return ::emu::from_bits<double>(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
inline long long __double_as_longlong(double v) { return ::emu::from_bits<long long>(::emu::to_bits(v)); }
inline double __longlong_as_double(long long v) { return ::emu::from_bits<double>(::emu::to_bits(v)); }
inline double rsqrt(double v) { return 1.0 / sqrt(v); }   // the device library's reciprocal square root
inline int __double2int_rz(double v) { return (int)v; }
inline int __double2int_rn(double v) { return (int)nearbyint(v); }
inline long long __double2ll_rz(double v) { return (long long)v; }
inline double __int2double_rn(int v) { return (double)v; }
inline int __ffsll(long long v) { return v ? __builtin_ctzll((unsigned long long)v) + 1 : 0; }
inline int __ffs(int v) { return v ? __builtin_ctz((unsigned)v) + 1 : 0; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

using std::max;
using std::min;
inline int64_t min(int64_t a, int b) { return a < b ? a : (int64_t)b; }
inline int64_t min(int a, int64_t b) { return a < b ? (int64_t)a : b; }
inline int64_t max(int64_t a, int b) { return a > b ? a : (int64_t)b; }
inline int64_t max(int a, int64_t b) { return a > b ? (int64_t)a : b; }

// blocks are interpreted on one host thread: the read-modify-write is trivially atomic
template <typename T, typename U> inline T atomicAdd(T* p, U v) { T old = *p; *p = old + (T)v; return old; }
template <typename T, typename U> inline T atomicMin(T* p, U v) { T old = *p; if ((T)v < old) *p = (T)v; return old; }
template <typename T, typename U> inline T atomicMax(T* p, U v) { T old = *p; if ((T)v > old) *p = (T)v; return old; }
template <typename T, typename U> inline T atomicExch(T* p, U v) { T old = *p; *p = (T)v; return old; }
template <typename T, typename U> inline T atomicOr(T* p, U v) { T old = *p; *p = old | (T)v; return old; }
inline void __threadfence() {}
inline void __threadfence_block() {}
