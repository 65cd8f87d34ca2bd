// common.hpp -- shared device/host helpers for libscint_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <type_traits>

#include "../../include/scint_hip.h"

namespace scint {

// ---- complex128 -------------------------------------------------------------
struct __attribute__((aligned(16))) cplx {
    double x, y;
};
static_assert(sizeof(cplx) == 16, "cplx must match numpy complex128");

__host__ __device__ inline cplx mk(double x, double y) { cplx r; r.x = x; r.y = y; return r; }
__host__ __device__ inline cplx operator+(cplx a, cplx b) { return mk(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplx operator-(cplx a, cplx b) { return mk(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplx operator*(cplx a, cplx b) {
    return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__host__ __device__ inline cplx operator*(cplx a, double s) { return mk(a.x * s, a.y * s); }
__host__ __device__ inline cplx conj(cplx a) { return mk(a.x, -a.y); }
// a * conj(b)
__host__ __device__ inline cplx mulc(cplx a, cplx b) {
    return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// multiply by -i (forward-FFT quarter turn) and by +i
__host__ __device__ inline cplx mul_mi(cplx a) { return mk(a.y, -a.x); }
__host__ __device__ inline cplx mul_pi(cplx a) { return mk(-a.y, a.x); }
__host__ __device__ inline double norm2(cplx a) { return a.x * a.x + a.y * a.y; }

// ---- explicit global-memory accesses ------------------------------------------------
// Pointers that are LOADED from a job table are generic to the compiler, which then emits
// flat_load/flat_store: those count on both vmcnt and lgkmcnt, so every wait becomes
// "wait for everything" and software pipelining is lost.  These helpers assert the global
// address space (global_load_dwordx4 / global_store_dwordx4, counted on vmcnt only).
typedef double v2d __attribute__((ext_vector_type(2)));
#define SCINT_GLOBAL __attribute__((address_space(1)))
__device__ inline cplx gload(const cplx* p) {
    const v2d v = *(const SCINT_GLOBAL v2d*)p;
    return mk(v.x, v.y);
}
// streamed-once data (the packed matrix): non-temporal hint keeps it from evicting the vectors
__device__ inline cplx gload_nt(const cplx* p) {
    const v2d v = __builtin_nontemporal_load((const SCINT_GLOBAL v2d*)p);
    return mk(v.x, v.y);
}
__device__ inline double gload(const double* p) { return *(const SCINT_GLOBAL double*)p; }
__device__ inline int32_t gload(const int32_t* p) { return *(const SCINT_GLOBAL int32_t*)p; }
__device__ inline uint8_t gload(const uint8_t* p) { return *(const SCINT_GLOBAL uint8_t*)p; }
__device__ inline void gstore(cplx* p, cplx v) {
    v2d t; t.x = v.x; t.y = v.y;
    *(SCINT_GLOBAL v2d*)p = t;
}
__device__ inline void gstore(double* p, double v) { *(SCINT_GLOBAL double*)p = v; }
__device__ inline void gstore_nt(cplx* p, cplx v) {
    v2d t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, (SCINT_GLOBAL v2d*)p);
}

// complex64 pair stored as two floats (8 bytes), non-temporal; and four floats (two complex64) loaded at once
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <class T>
__device__ inline void gstore_nt(T* p, double re, double im) {
    static_assert(sizeof(T) == 8, "complex64 element");
    v2f t; t.x = (float)re; t.y = (float)im;
    __builtin_nontemporal_store(t, (SCINT_GLOBAL v2f*)p);
}
template <class T>
__device__ inline v4f gload_nt4(const T* p) {
    static_assert(sizeof(T) == 8, "complex64 element");
    return __builtin_nontemporal_load((const SCINT_GLOBAL v4f*)p);
}

// ---- buffer-resource accesses -------------------------------------------------------------------------
// An array behind a 128-bit resource descriptor (SGPRs): a load is then base + 32-bit per-lane byte offset (ONE
// register) + scalar byte offset, with a hardware bounds check (out of range reads 0).  In a persistent loop the
// compiler turns sixteen global loads at constant distances into sixteen loop-carried 64-bit addresses (and spills
// them); here the sixteen distances are scalar operands.  The array must be smaller than 4 GiB.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ inline BufRsrc make_rsrc(const void* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ inline cplx bload_c(BufRsrc r, int voff, int soff) {
    const v2d v = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return mk(v.x, v.y);
}
__device__ inline double bload_d(BufRsrc r, int voff, int soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ inline void bstore2(BufRsrc r, int voff, int soff, double x, double y) {
    v2d v; v.x = x; v.y = y;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, voff, soff, 0);
}

// ---- workgroup barrier for LDS hand-offs ---------------------------------------------------------
// __syncthreads() is a workgroup-scope fence + barrier: the fence also waits for every GLOBAL access
// in flight (s_waitcnt vmcnt(0)), i.e. it drains the loads a kernel has prefetched for its next
// tile.  Where the only thing handed from wave to wave is LDS data, waiting for the LDS counter is
// enough.  (tests/emu turns this line into a plain barrier.)
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Lanes of ONE wavefront exchanging data through LDS: the LDS executes a wave's instructions in order,
// so the hardware needs no barrier; this only pins the order for the compiler (and is a wave-level
// meeting point on the host interpreter of tests/emu).
__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Lane K of every 16-lane row to all lanes of that row (v_mov_b32_dpp row_newbcast:K, VALU only)
template <int K>
__device__ inline double row_bcast_f64(double v) {
    static_assert(K >= 0 && K < 16, "row_newbcast lane");
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (DPP controls are immediates)
template <int B, int E, class F>
__device__ inline void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- wavefront (64 lanes) reductions ------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline cplx wave_sum(cplx v) { return mk(wave_sum(v.x), wave_sum(v.y)); }

// Deterministic block reduction (fixed tree): every thread gets the total.
// `red` must hold blockDim.x/64 entries.
template <typename T>
__device__ inline T block_sum(T v, T* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    T t = red[0];
    for (int i = 1; i < nw; ++i) t = t + red[i];
    return t;
}

// ---- error plumbing -------------------------------------------------------------
void set_error(const std::string& msg);
int32_t hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SCINT_HIP(call)                                                         \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return ::scint::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define SCINT_LAUNCH_CHECK() SCINT_HIP(hipGetLastError())

#define SCINT_REQUIRE(cond, msg)                                \
    do {                                                        \
        if (!(cond)) {                                          \
            ::scint::set_error(std::string("scint: ") + (msg)); \
            return SCINT_E_ARG;                                 \
        }                                                       \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }
inline int ilog2(int64_t n) { int l = 0; while ((int64_t(1) << l) < n) ++l; return l; }
inline int64_t next_pow2(int64_t n) { return int64_t(1) << ilog2(n); }

// Carve aligned sub-buffers out of the caller's workspace.
struct Carver {
    char* base; size_t off; size_t cap;
    Carver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes) {}
    template <typename T> T* take(size_t count) {
        off = align_up(off, 256);
        T* r = (T*)(base + off);
        off += count * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

}  // namespace scint
