// TEST INFRASTRUCTURE ONLY (see include/hip/hip_runtime.h in this directory).
//
// A workgroup interpreter for the x86 host: one fiber per GPU thread, 64 consecutive fibers form
// a wavefront.  A fiber runs until it reaches a scheduling point -- a block barrier, a cross-lane
// operation, or the end of the kernel.  Cross-lane operations are resolved when every lane of the
// wave is parked (at a cross-lane operation, at a barrier, or finished): the lanes parked at the
// operation are the active lanes, exactly the set the hardware would execute it for under
// structured control flow.  A barrier releases when every fiber of the block is parked at it or
// finished.  Blocks run one after the other, kernels run synchronously at launch.
// One host thread at a time: the fiber table and the geometry variables are process-global (the
// tests drive the library from one thread; ranks of a multi-process test are separate processes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

// dynamic LDS of the kernels (declared `extern __shared__` in the sources; the build script turns
// that into a plain extern of these arrays)
namespace scint {
alignas(64) char smem_raw[192 * 1024];
alignas(64) double rev_lds[192 * 1024 / 8];
}  // namespace scint

extern "C" void emu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n.globl emu_switch\n.type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_switch,.-emu_switch\n");

namespace emu {
namespace {
enum State { kRun = 0, kAtBarrier, kAtWave, kDone };
constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = kDone;
    dim3 tid;
    int op = 0, arg = 0, width = 64;
    uint64_t value = 0, value2 = 0, result = 0;
};
constexpr int kGather = 100;
uint64_t snapshots[kMaxThreads / 64][64][2];
Fiber fibers[kMaxThreads];
void* sched_sp = nullptr;
int cur = -1;
void (*g_thunk)(void*) = nullptr;
void* g_ctx = nullptr;

void to_scheduler() { emu_switch(&fibers[cur].sp, sched_sp); }

void fiber_main() {
    g_thunk(g_ctx);
    fibers[cur].state = kDone;
    to_scheduler();
    abort();   // a finished fiber is never resumed
}

void prepare(Fiber& f) {
    if (!f.stack) {
        if (posix_memalign((void**)&f.stack, 64, kStackBytes) != 0) abort();
    }
    uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)63;
    void** sp = (void**)(top - 128);          // 16-byte aligned
    // layout popped by emu_switch: r15 r14 r13 r12 rbx rbp, then `ret` into fiber_main
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    sp[6] = (void*)&fiber_main;
    sp[7] = nullptr;                          // fake return address of fiber_main
    f.sp = sp;
    f.state = kRun;
}

void resume(int i) {
    cur = i;
    threadIdx = fibers[i].tid;
    emu_switch(&sched_sp, fibers[i].sp);
    cur = -1;
}

void resolve_wave(int lo, int hi) {
    // first pass: results from the parked values; second pass: release
    int first = -1;
    uint64_t ballot = 0;
    int op0 = 0;
    for (int l = lo; l < hi; ++l) {
        if (fibers[l].state != kAtWave) continue;
        if (first < 0) { first = l; op0 = fibers[l].op; }
        if (fibers[l].op != op0) {
            fprintf(stderr, "emu: lanes of one wave parked at different cross-lane operations (%d vs %d): "
                            "divergent control flow around a wave operation\n", op0, fibers[l].op);
            abort();
        }
        if (fibers[l].value) ballot |= 1ull << (l - lo);
    }
    if (op0 == kGather) {
        uint64_t (*snap)[2] = snapshots[lo / 64];
        for (int l = lo; l < lo + 64; ++l) {
            const bool on = l < hi && fibers[l].state == kAtWave;
            snap[l - lo][0] = on ? fibers[l].value : 0;
            snap[l - lo][1] = on ? fibers[l].value2 : 0;
            if (on) fibers[l].state = kRun;
        }
        return;
    }
    for (int l = lo; l < hi; ++l) {
        Fiber& f = fibers[l];
        if (f.state != kAtWave) continue;
        const int lane = l - lo, w = f.width > 0 && f.width <= 64 ? f.width : 64;
        const int base = lane & ~(w - 1);
        int src = lane;
        switch (f.op) {
            case kShfl: src = base + (f.arg & (w - 1)); break;
            case kShflXor: src = lane ^ f.arg; if (src < base || src >= base + w) src = lane; break;
            case kShflDown: src = lane + f.arg; if (src >= base + w) src = lane; break;
            case kShflUp: src = lane - f.arg; if (src < base) src = lane; break;
            case kBallot: f.result = ballot; continue;
            case kReadFirst: f.result = fibers[first].value; continue;
            default: abort();
        }
        const int s = lo + src;
        f.result = (s < hi && fibers[s].state == kAtWave) ? fibers[s].value : f.value;
    }
    for (int l = lo; l < hi; ++l)
        if (fibers[l].state == kAtWave) fibers[l].state = kRun;
}

// SCINT_EMU_ORDER=rev runs the waves of a block, the lanes of a wave and the blocks of a grid in
// the opposite order.  Every order is a legal schedule, so results must not depend on it: a
// difference means a missing barrier, an inter-block dependence, or reliance on lock-step lanes.
bool reversed_order() {
    static const bool rev = [] { const char* e = getenv("SCINT_EMU_ORDER"); return e && e[0] == 'r'; }();
    return rev;
}

void run_block(int nthreads) {
    const bool rev = reversed_order();
    const int nwaves = (nthreads + 63) / 64;
    for (;;) {
        for (int wi = 0; wi < nwaves; ++wi) {
            const int w0 = 64 * (rev ? nwaves - 1 - wi : wi);
            const int w1 = std::min(nthreads, w0 + 64);
            for (;;) {
                bool parked_at_wave = false;
                for (int li = 0; li < w1 - w0; ++li) {
                    const int l = rev ? w1 - 1 - li : w0 + li;
                    if (fibers[l].state == kRun) resume(l);
                    if (fibers[l].state == kAtWave) parked_at_wave = true;
                }
                if (!parked_at_wave) break;
                resolve_wave(w0, w1);
            }
        }
        bool any = false;
        for (int l = 0; l < nthreads; ++l)
            if (fibers[l].state == kAtBarrier) { fibers[l].state = kRun; any = true; }
        if (!any) return;
    }
}
}  // namespace

uint64_t wave_op(int op, uint64_t value, int arg, int width) {
    Fiber& f = fibers[cur];
    f.op = op; f.value = value; f.arg = arg; f.width = width;
    f.state = kAtWave;
    to_scheduler();
    return f.result;
}

const uint64_t (*wave_gather2(uint64_t v0, uint64_t v1))[2] {
    Fiber& f = fibers[cur];
    f.op = kGather; f.value = v0; f.value2 = v1;
    f.state = kAtWave;
    const int wave = cur / 64;
    to_scheduler();
    return snapshots[wave];
}

int lane_id() { return cur & 63; }

void block_barrier() {
    fibers[cur].state = kAtBarrier;
    to_scheduler();
}

void run_grid(dim3 grid, dim3 block, size_t shmem, void (*thunk)(void*), void* ctx) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > kMaxThreads || shmem > sizeof(scint::smem_raw)) {
        fprintf(stderr, "emu: unsupported launch (%d threads, %zu bytes of dynamic LDS)\n", nthreads, shmem);
        abort();
    }
    if (cur >= 0) { fprintf(stderr, "emu: nested launch\n"); abort(); }
    g_thunk = thunk; g_ctx = ctx;
    gridDim = grid; blockDim = block;
    const bool rev = reversed_order();
    for (uint32_t iz = 0; iz < grid.z; ++iz)
        for (uint32_t iy = 0; iy < grid.y; ++iy)
            for (uint32_t ix = 0; ix < grid.x; ++ix) {
                const uint32_t bx = rev ? grid.x - 1 - ix : ix, by = rev ? grid.y - 1 - iy : iy,
                               bz = rev ? grid.z - 1 - iz : iz;
                blockIdx = dim3(bx, by, bz);
                int t = 0;
                for (uint32_t z = 0; z < block.z; ++z)
                    for (uint32_t y = 0; y < block.y; ++y)
                        for (uint32_t x = 0; x < block.x; ++x, ++t) {
                            fibers[t].tid = dim3(x, y, z);
                            prepare(fibers[t]);
                        }
                run_block(nthreads);
            }
}
}  // namespace emu

// ---- runtime API: device memory is host memory, streams are synchronous ---------------------------
struct emu_stream { int id; };
struct emu_event { double t_ms; };

static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t bytes) {
    return posix_memalign(p, 256, bytes ? bytes : 1) == 0 ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) {
    memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { memset(dst, value, bytes); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emu_stream{1}; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0.0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t_ms = now_ms(); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
