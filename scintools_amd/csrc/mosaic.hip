// mosaic.hip -- the tail of phase retrieval on the device (ththmod.py:1492-1554 `mosaic`; dynspec.py:1765-1826 the chunk loop
// of Dynspec.thetatheta_chunks): the half-overlapping wavefield chunks stay in HBM, each is rotated onto what is already there
// and added under its taper; and the chunks themselves are cut out of the dynamic spectrum on the device.
//
// The reference's loop is sequentially dependent (the phase of chunk k is taken against the sum of chunks 0..k-1) and its
// arithmetic is NumPy's: `(chunk_old * conj(chunk_new) * mask).mean()` is a sum in NumPy's own order.  That order is restated
// here exactly, so that the device mosaic equals the host loop bit for bit (tests/test_emu_cpu.py, tests/test_gpu_parity.py):
//
//   numpy.add.reduce over a contiguous array walks it in pieces of the ufunc buffer (8192 ELEMENTS), adds each piece's PAIRWISE
//   sum to a running total, and a pairwise sum (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum) is: up to 128
//   doubles -> eight strided accumulators r[j] += a[i + j], combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) -- for complex data
//   (re, im interleaved) (r0+r2)+(r4+r6) and (r1+r3)+(r5+r7) --, then the <8 leftover values one by one; longer runs split at
//   n/2 rounded down to a multiple of 8 and add the two halves' sums.
//
// A `PairwisePlan` is that recursion unrolled on the host for one element count: the leaves (runs of <= 128 doubles, one thread
// each), the additions level by level (parallel within a level), the piece roots in order (one thread).  The scalar steps between
// the sum and the update -- the division by the count, numpy.angle, numpy.exp -- stay in Python, in NumPy itself (its atan2 / sin /
// cos need not be the C library's): the sum travels to the host (16 bytes), the phase factor comes back as two kernel arguments.
//
// Compiled with -ffp-contract=off (scintools_amd/build.py): a fused multiply-add rounds differently from NumPy's separate operations.
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "thth.hpp"

namespace scint {

constexpr int kPwBlock = 128;        // numpy PW_BLOCKSIZE (doubles)
constexpr int kNpBuffer = 8192;      // numpy.getbufsize(): elements per inner-loop call of a reduction

struct PwLeaf { int32_t start, len; };             // in doubles, relative to the array
struct PwOp { int32_t dst, a, b; };                 // node[dst] = node[a] + node[b]
struct PairwisePlanDev {
    const PwLeaf* leaves; int nleaves;
    const PwOp* ops; const int32_t* level_off; int nlevels;       // ops[level_off[l] .. level_off[l+1]) are independent
    const int32_t* roots; int nroots;                              // total = ((0 + node[roots[0]]) + node[roots[1]]) + ...
    int nnodes;
};

static std::mutex g_plan_mutex;
static std::map<std::tuple<int, int64_t, int>, PairwisePlanDev> g_plan_cache;

template <class T>
static T* upload_vec(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(T) * std::max<size_t>(v.size(), 1)) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

// numpy's recursion as a tree: a node is a leaf (a run of <= 128 doubles) or the sum of its two children
struct PwTree {
    struct Node { int left, right, leaf, height; };
    std::vector<Node> nodes;
    std::vector<PwLeaf> leaves;
    int build(int64_t lo, int64_t n) {
        if (n <= kPwBlock) {
            leaves.push_back(PwLeaf{(int32_t)lo, (int32_t)n});
            nodes.push_back(Node{-1, -1, (int)leaves.size() - 1, 0});
            return (int)nodes.size() - 1;
        }
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        const int l = build(lo, n2), r = build(lo + n2, n - n2);
        nodes.push_back(Node{l, r, -1, std::max(nodes[(size_t)l].height, nodes[(size_t)r].height) + 1});
        return (int)nodes.size() - 1;
    }
};

// The plan of numpy.add.reduce over `count` contiguous elements of `width` doubles each (cached per device for the life of the
// process, like the FFT twiddle tables)
static const PairwisePlanDev* pairwise_plan(int64_t count, int width) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("scint: hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    const auto key = std::make_tuple(dev, count, width);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) return &it->second;
    PwTree t;
    std::vector<int> piece_roots;
    for (int64_t e0 = 0; e0 < count; e0 += kNpBuffer)                 // one pairwise sum per buffer piece
        piece_roots.push_back(t.build(e0 * width, std::min<int64_t>(kNpBuffer, count - e0) * width));
    const int nleaves = (int)t.leaves.size();
    int hmax = 0;
    for (const auto& nd : t.nodes) hmax = std::max(hmax, nd.height);
    // device numbering: the leaves first (in order), then the sums level by level (a sum's children are on lower levels)
    std::vector<int> dev_index(t.nodes.size(), -1);
    std::vector<int32_t> level_off((size_t)hmax + 1, 0);
    int next = nleaves;
    for (size_t k = 0; k < t.nodes.size(); ++k)
        if (t.nodes[k].leaf >= 0) dev_index[k] = t.nodes[k].leaf;
    std::vector<PwOp> ops;
    for (int h = 1; h <= hmax; ++h) {
        level_off[(size_t)h - 1] = (int32_t)ops.size();
        for (size_t k = 0; k < t.nodes.size(); ++k)
            if (t.nodes[k].height == h) {
                dev_index[k] = next++;
                ops.push_back(PwOp{dev_index[k], dev_index[(size_t)t.nodes[k].left], dev_index[(size_t)t.nodes[k].right]});
            }
    }
    level_off[(size_t)hmax] = (int32_t)ops.size();
    std::vector<int32_t> roots;
    for (int r : piece_roots) roots.push_back(dev_index[(size_t)r]);
    PairwisePlanDev p{};
    p.leaves = upload_vec(t.leaves); p.nleaves = nleaves;
    p.ops = upload_vec(ops); p.level_off = upload_vec(level_off); p.nlevels = hmax;
    p.roots = upload_vec(roots); p.nroots = (int)roots.size();
    p.nnodes = next;
    if (!p.leaves || !p.ops || !p.level_off || !p.roots) { set_error("scint: pairwise plan: device allocation failed"); return nullptr; }
    return &(g_plan_cache[key] = p);
}

// One leaf of a pairwise sum: `len` doubles produced by `val(i)` (i-th double of the leaf), numpy's eight accumulators.
// CPLX: the doubles alternate (re, im); rr / ri are the two sums (real data: rr only).
template <bool CPLX, class F>
__device__ inline void pw_leaf(int len, F&& val, double& rr, double& ri) {
    if (len < 8) {
        rr = 0.0; ri = 0.0;                        // (numpy >= 1.24 starts from -0.0; the pieces of this library are never that short)
        if (CPLX) { for (int i = 0; i < len; i += 2) { rr = rr + val(i); ri = ri + val(i + 1); } }
        else { for (int i = 0; i < len; ++i) rr = rr + val(i); }
        return;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = val(j);
    int i = 8;
    for (; i < len - (len % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + val(i + j);
    }
    if (CPLX) {
        rr = (r[0] + r[2]) + (r[4] + r[6]);
        ri = (r[1] + r[3]) + (r[5] + r[7]);
        for (; i < len; i += 2) { rr = rr + val(i); ri = ri + val(i + 1); }
    } else {
        rr = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        ri = 0.0;
        for (; i < len; ++i) rr = rr + val(i);
    }
}

// the additions of a plan, level by level, then the pieces in order; nodes in global memory (one workgroup: a barrier orders them)
__device__ inline void pw_combine(const PairwisePlanDev& p, double* node_re, double* node_im, bool cplx, double* out) {
    for (int l = 0; l < p.nlevels; ++l) {
        __syncthreads();
        for (int k = p.level_off[l] + (int)threadIdx.x; k < p.level_off[l + 1]; k += (int)blockDim.x) {
            const PwOp op = p.ops[k];
            node_re[op.dst] = node_re[op.a] + node_re[op.b];
            if (cplx) node_im[op.dst] = node_im[op.a] + node_im[op.b];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < p.nroots; ++k) {
            sr = sr + node_re[p.roots[k]];
            if (cplx) si = si + node_im[p.roots[k]];
        }
        out[0] = sr;
        if (cplx) out[1] = si;
    }
}

// One chunk of a step of the mosaic: where its window starts in the wavefield, which chunk of the stack, which of the four
// row / column tapers (2 * has a neighbour before + has a neighbour after).  int64 [4] per job in device memory.
struct MosaicArgs {
    cplx* E; int64_t ldE;                 // the wavefield so far, [F][T]
    const cplx* chunks; int cwf, cwt;     // the stack [nchunk][cwf][cwt]
    const int64_t* jobs;                  // [.][4]: window offset into E, chunk index, row taper, column taper; this launch's first job
    const double* rows; const double* cols;   // tapers [4][cwf], [4][cwt]: mask[r][c] = rows[rv][r] * cols[cv][c]  (ththmod.py:1526-1546)
    int fused;                            // how the HOST's numpy evaluates the products (the wrapper measures it): bit 0 fused multiply-adds
                                          // (cmul_np), bit 1 operands of chunk_old * conj(chunk_new) swapped (temporary elision)
};
constexpr int kMosaicStep = 64;           // most chunks per launch (their phase factors travel as kernel arguments)
struct MosaicPhases { double re[kMosaicStep], im[kMosaicStep]; };
// numpy's product of two complex128 ARRAY elements a * b.  Its SIMD loops (x86 with FMA3: AVX2 / AVX-512 dispatch) compute
// re = fma(ar, br, -(ai bi)), im = fma(ar, bi, ai br) -- one rounding fewer than the plain expressions its scalar loop uses.
// Which one a host runs is a property of that host's CPU; the wrapper measures it once (ththmod._numpy_complex_product_is_fused)
// and hands the answer down, so that the device mosaic equals the host loop bit for bit on either kind.
__device__ inline cplx cmul_np(cplx a, cplx b, bool fused) {
    if (fused) return mk(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
    return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// per job of the launch: sum over the window of  (E * conj(chunk)) * mask  in numpy's order  ->  out[job][0..1].  Two kernels: the leaves
// (blockIdx.y = job, kLeafBlocks workgroups of 256 threads share its leaves: one workgroup per chunk took 238 us a step, a CU each
// for at most sixteen chunks), then the additions (one workgroup per job).
constexpr int kLeafBlocks = 8;
__global__ void __launch_bounds__(256) mosaic_phase_leaves_kernel(MosaicArgs a, PairwisePlanDev p, double* nodes) {
    const int64_t* job = a.jobs + 4 * (int64_t)blockIdx.y;
    const cplx* __restrict__ E = a.E + job[0];
    const cplx* __restrict__ chunk = a.chunks + job[1] * (int64_t)a.cwf * a.cwt;
    const double* __restrict__ fr = a.rows + job[2] * a.cwf;
    const double* __restrict__ fc = a.cols + job[3] * a.cwt;
    double* node_re = nodes + (size_t)blockIdx.y * 2 * (size_t)p.nnodes;
    double* node_im = node_re + p.nnodes;
    for (int lf = (int)(blockIdx.x * 256 + threadIdx.x); lf < p.nleaves; lf += (int)gridDim.x * 256) {
        const PwLeaf L = p.leaves[lf];
        auto val = [&](int i) {
            const int d = L.start + i, e = d >> 1;             // element of the window, row-major
            const int r = e / a.cwt, c = e - r * a.cwt;
            const cplx o = E[(int64_t)r * a.ldE + c], n = chunk[e];
            // chunk_old * numpy.conjugate(chunk_new): for temporaries of 256 KiB and more numpy stores the product in the
            // conjugate's buffer and, the operation being commutative, evaluates conj(chunk_new) * chunk_old (bit 1 of `fused`)
            const cplx cn = mk(n.x, -n.y);
            const cplx t = (a.fused & 2) ? cmul_np(cn, o, (a.fused & 1) != 0) : cmul_np(o, cn, (a.fused & 1) != 0);
            const double m = fr[r] * fc[c];                    // (the real mask: numpy multiplies by m + 0j, i.e. both parts by m)
            return (d & 1) ? t.y * m : t.x * m;
        };
        double rr, ri;
        pw_leaf<true>(L.len, val, rr, ri);
        node_re[lf] = rr; node_im[lf] = ri;
    }
}
__global__ void __launch_bounds__(256) mosaic_phase_combine_kernel(PairwisePlanDev p, double* nodes, double* out) {
    double* node_re = nodes + (size_t)blockIdx.x * 2 * (size_t)p.nnodes;
    pw_combine(p, node_re, node_re + p.nnodes, true, out + 2 * (size_t)blockIdx.x);
}

// E[window] += (chunk * mask) * phase[job]      (ththmod.py:1551: E_recov[...] += chunk_new * mask * exp(1j * rot)); blockIdx.y = job
__global__ void __launch_bounds__(256) mosaic_add_kernel(MosaicArgs a, MosaicPhases ph) {
    const int64_t* job = a.jobs + 4 * (int64_t)blockIdx.y;
    cplx* __restrict__ E = a.E + job[0];
    const cplx* __restrict__ chunk = a.chunks + job[1] * (int64_t)a.cwf * a.cwt;
    const double* __restrict__ fr = a.rows + job[2] * a.cwf;
    const double* __restrict__ fc = a.cols + job[3] * a.cwt;
    const cplx phase = mk(ph.re[blockIdx.y], ph.im[blockIdx.y]);
    const int n = a.cwf * a.cwt;
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < n; e += (int)(gridDim.x * blockDim.x)) {
        const int r = e / a.cwt, c = e - r * a.cwt;
        const cplx v = chunk[e];
        const double m = fr[r] * fc[c];
        const cplx y = cmul_np(mk(v.x * m, v.y * m), phase, (a.fused & 1) != 0);
        cplx* dst = E + (int64_t)r * a.ldE + c;
        const cplx o = *dst;
        *dst = mk(o.x + y.x, o.y + y.y);
    }
}

// ---- chunks cut out of the dynamic spectrum (dynspec.py:1782-1790: copy, subtract numpy.nanmean, numpy.nan_to_num) ------------
// One workgroup per chunk.  nanmean = sum of the non-NaN values (numpy sums a copy with the NaNs replaced by 0: the same
// pairwise order) / their count; the padding value of the conjugate spectrum is the mean of the RESULT (ththmod.py:783).
struct CutArgs {
    const double* dyn; int64_t ld;        // [nf_all][ld]
    const int32_t* r0; const int32_t* c0; // window origin of every chunk
    int cwf, cwt;
    double* out;                          // [nchunk][cwf][cwt]
    double* pad;                          // [nchunk]: mean of the processed chunk
    int colmajor;                         // the HOST array is Fortran-ordered (a transposed view, as psrflux files load): numpy.copy keeps
                                          // that order and its sums walk the window column by column -- so do these (the data here are row-major)
};
// element e of the window in the order numpy's sums visit it -> (row, column)
__device__ inline void cut_rc(const CutArgs& a, int e, int& r, int& c) {
    if (a.colmajor) { c = e / a.cwf; r = e - c * a.cwf; }
    else { r = e / a.cwt; c = e - r * a.cwt; }
}
__device__ inline double nan_to_num_f64(double v) {
    if (v != v) return 0.0;
    if (v == INFINITY) return 1.7976931348623157e308;
    if (v == -INFINITY) return -1.7976931348623157e308;
    return v;
}
__global__ void __launch_bounds__(1024) chunk_cut_kernel(CutArgs a, PairwisePlanDev p, double* node_all) {
    __shared__ double s_sum[2];
    __shared__ int s_cnt[32];
    const int k = (int)blockIdx.x, n = a.cwf * a.cwt;
    const double* __restrict__ src = a.dyn + (int64_t)a.r0[k] * a.ld + a.c0[k];
    double* node = node_all + (size_t)k * (size_t)p.nnodes;
    double* __restrict__ dst = a.out + (size_t)k * (size_t)n;
    // (1) nanmean: pairwise sum of where(isnan, 0, x), and the count of the finite-or-infinite values
    int cnt = 0;
    for (int lf = (int)threadIdx.x; lf < p.nleaves; lf += (int)blockDim.x) {
        const PwLeaf L = p.leaves[lf];
        auto val = [&](int i) {
            int r, c;
            cut_rc(a, L.start + i, r, c);
            const double v = src[(int64_t)r * a.ld + c];
            return (v != v) ? 0.0 : v;
        };
        double rr, ri;
        pw_leaf<false>(L.len, val, rr, ri);
        node[lf] = rr;
        for (int i = 0; i < L.len; ++i) {
            int r, c;
            cut_rc(a, L.start + i, r, c);
            const double v = src[(int64_t)r * a.ld + c];
            cnt += (v == v) ? 1 : 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    pw_combine(p, node, nullptr, false, s_sum);
    __syncthreads();
    int total = 0;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) total += s_cnt[w];
    const double mean = s_sum[0] / (double)total;                       // (no finite value: 0 / 0 = NaN, as numpy.nanmean)
    // (2) the chunk: nan_to_num(x - mean)
    for (int e = (int)threadIdx.x; e < n; e += (int)blockDim.x) {
        const int r = e / a.cwt, c = e - r * a.cwt;
        dst[e] = nan_to_num_f64(src[(int64_t)r * a.ld + c] - mean);
    }
    __syncthreads();
    // (3) its mean (the padding value): numpy.mean = add.reduce / count
    for (int lf = (int)threadIdx.x; lf < p.nleaves; lf += (int)blockDim.x) {
        const PwLeaf L = p.leaves[lf];
        auto val = [&](int i) { int r, c; cut_rc(a, L.start + i, r, c); return dst[r * a.cwt + c]; };
        double rr, ri;
        pw_leaf<false>(L.len, val, rr, ri);
        node[lf] = rr;
    }
    pw_combine(p, node, nullptr, false, s_sum + 1);
    __syncthreads();
    if (threadIdx.x == 0) a.pad[k] = s_sum[1] / (double)n;
}

}  // namespace scint

using namespace scint;

extern "C" int32_t scint_mosaic_workspace_bytes(int64_t cwf, int64_t cwt, int64_t nchunk, size_t* bytes) {
    SCINT_REQUIRE(bytes != nullptr, "mosaic_workspace_bytes: null output");
    SCINT_REQUIRE(cwf >= 1 && cwt >= 1 && nchunk >= 1 && cwf * cwt < (int64_t(1) << 29), "mosaic_workspace_bytes: bad shape");
    // nodes of a plan: fewer than twice its leaves; leaves: at most ceil(2 n / 64) + one short one per buffer piece
    const int64_t n = cwf * cwt, leaves = (2 * n + 63) / 64 + n / kNpBuffer + 2;
    *bytes = (size_t)(2 * leaves) * 2 * sizeof(double) * (size_t)nchunk + 256;      // (real and imaginary node sums per chunk)
    return SCINT_OK;
}

extern "C" int32_t scint_mosaic_phase(const scint_c128* E, int64_t ldE, const scint_c128* chunks, int64_t cwf, int64_t cwt,
                                      const int64_t* jobs, int64_t count, const double* rows, const double* cols,
                                      int32_t numpy_fused, void* workspace, size_t workspace_bytes, double* sums_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SCINT_REQUIRE(E && chunks && jobs && rows && cols && workspace && sums_out, "mosaic_phase: null pointer");
    SCINT_REQUIRE(cwf >= 1 && cwt >= 1 && ldE >= cwt && count >= 1 && count <= kMosaicStep, "mosaic_phase: bad shape or more than 64 chunks");
    size_t need = 0;
    if (scint_mosaic_workspace_bytes(cwf, cwt, count, &need) != SCINT_OK) return SCINT_E_ARG;
    if (workspace_bytes < need) { set_error("scint: mosaic workspace too small"); return SCINT_E_WORKSPACE; }
    const PairwisePlanDev* p = pairwise_plan(cwf * cwt, 2);
    if (!p) return SCINT_E_HIP;
    MosaicArgs a{(cplx*)E, ldE, (const cplx*)chunks, (int)cwf, (int)cwt, jobs, rows, cols, numpy_fused};
    hipLaunchKernelGGL(mosaic_phase_leaves_kernel, dim3(kLeafBlocks, (unsigned)count), dim3(256), 0, stream, a, *p, (double*)workspace);
    hipLaunchKernelGGL(mosaic_phase_combine_kernel, dim3((unsigned)count), dim3(256), 0, stream, *p, (double*)workspace, sums_out);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

extern "C" int32_t scint_mosaic_add(scint_c128* E, int64_t ldE, const scint_c128* chunks, int64_t cwf, int64_t cwt,
                                    const int64_t* jobs, int64_t count, const double* rows, const double* cols,
                                    int32_t numpy_fused, const double* phases /*HOST [count][2]*/, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SCINT_REQUIRE(E && chunks && jobs && rows && cols && phases, "mosaic_add: null pointer");
    SCINT_REQUIRE(cwf >= 1 && cwt >= 1 && ldE >= cwt && cwf * cwt < (int64_t(1) << 30) && count >= 1 && count <= kMosaicStep,
                  "mosaic_add: bad shape or more than 64 chunks");
    MosaicArgs a{(cplx*)E, ldE, (const cplx*)chunks, (int)cwf, (int)cwt, jobs, rows, cols, numpy_fused};
    MosaicPhases ph;
    for (int64_t k = 0; k < count; ++k) { ph.re[k] = phases[2 * k]; ph.im[k] = phases[2 * k + 1]; }
    const int blocks = (int)std::min<int64_t>(ceil_div(cwf * cwt, 256), 256);
    hipLaunchKernelGGL(mosaic_add_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, stream, a, ph);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

extern "C" int32_t scint_chunk_cut(const double* dyn, int64_t nf, int64_t nt, const int32_t* r0, const int32_t* c0, int64_t nchunk,
                                   int64_t cwf, int64_t cwt, int32_t colmajor, double* chunks_out, double* pad_out, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SCINT_REQUIRE(dyn && r0 && c0 && chunks_out && pad_out && workspace, "chunk_cut: null pointer");
    SCINT_REQUIRE(nchunk >= 1 && cwf >= 1 && cwt >= 1 && cwf <= nf && cwt <= nt, "chunk_cut: bad shape");
    size_t need = 0;
    if (scint_mosaic_workspace_bytes(cwf, cwt, nchunk, &need) != SCINT_OK) return SCINT_E_ARG;
    if (workspace_bytes < need) { set_error("scint: chunk_cut workspace too small"); return SCINT_E_WORKSPACE; }
    const PairwisePlanDev* p = pairwise_plan(cwf * cwt, 1);
    if (!p) return SCINT_E_HIP;
    CutArgs a{dyn, nt, r0, c0, (int)cwf, (int)cwt, chunks_out, pad_out, colmajor ? 1 : 0};
    hipLaunchKernelGGL(chunk_cut_kernel, dim3((unsigned)nchunk), dim3(1024), 0, stream, a, *p, (double*)workspace);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

// ---- the per-chunk steps of Dynspec.thetatheta_chunks in ONE call each (dynspec.py:1765-1826 maps single_chunk_retrieval over the
// chunks; 961 of them for a 4096^2 observation): the same kernels as scint_cs / scint_rev_map / scint_ifft2_shifted, queued by a
// C++ loop instead of a dozen Python calls per chunk (0.17 s of a 0.31-s calc_wavefield was that: profiles/r06_wavefield_host_profile_after.txt).
extern "C" int32_t scint_cs_batch(const double* dstack, int64_t n, int64_t nf, int64_t nt, int64_t npad, const double* pads,
                                  const int64_t* mask_lohi, int32_t incoherent, scint_c128* cs_stack, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    SCINT_REQUIRE(dstack && pads && mask_lohi && cs_stack && workspace && n >= 1, "cs_batch: null pointer or empty batch");
    const int64_t R = (npad + 1) * nf, C = (npad + 1) * nt;
    for (int64_t k = 0; k < n; ++k) {
        const int32_t rc = scint_cs(dstack + k * nf * nt, nf, nt, npad, pads[k], mask_lohi[2 * k], mask_lohi[2 * k + 1], incoherent,
                                    cs_stack + k * R * C, workspace, workspace_bytes, stream);
        if (rc != SCINT_OK) return rc;
    }
    return SCINT_OK;
}

// ---- the back-map of single_chunk_retrieval without its N x N matrix -------------------------------------------------------
// ththmod.py:1457-1464 maps a theta-theta matrix that is zero except for ONE row (row N/2 = conj(V) sqrt(w)) with
// rev_map(hermetian=False): every pixel is (sum of the weights that fall in it) / (number of ALL N^2 theta-theta pixels that
// fall in it).  The counts depend on the chunk's theta grid, curvature and axes only -- the same for every chunk of one frequency
// row of the observation -- so they are formed once per such class (integer atomics: exact), and a chunk costs N pairs instead
// of N^2 and no zero matrix.  A pixel's weights are added in increasing j, NumPy's order (bincount adds in input order).
namespace scint {
__global__ void __launch_bounds__(256) rev_count_kernel(const double* __restrict__ th, int N, double eta, GeomDev g, uint32_t* __restrict__ cnt) {
    const int64_t total = (int64_t)N * N;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int i = (int)(q / N), j = (int)(q - (int64_t)i * N);
        if (i == j) continue;                                          // (the centre bin these poison is written as 0: below)
        const double ti = th[i], tj = th[j];
        const int64_t bx = hist_bin(tj - ti, g.fd0, g.fd1_step, g.nfd);             // fd_map[i, j]   (ththmod.py:207)
        const int64_t by = hist_bin(eta * (tj * tj - ti * ti), g.tau0, g.tau1_step, g.ntau);   // tau_map[i, j] (:208-210)
        if (bx >= 0 && by >= 0) atomicAdd(cnt + by * g.nfd + bx, 1u);
    }
}
constexpr int kRowRun = 8;     // positions looked at either side for the same pixel (a Doppler bin holds two or three centres)
constexpr int kTailGroup = 8;  // chunks of a class whose back-maps and zero-fills go in one launch (their images side by side)
struct TailMembers { int32_t k[kTailGroup]; };
// one workgroup per chunk of the class: pixel and weight of every j, then each pixel's first j adds its run and writes it
__global__ void __launch_bounds__(256) rev_row_kernel(const cplx* __restrict__ rows, const double* __restrict__ th_red, int64_t M,
                                                      TailMembers members, int N, double eta, GeomDev g,
                                                      const uint32_t* __restrict__ cnt, cplx* __restrict__ recov_all, int64_t centre,
                                                      int64_t* __restrict__ pix_all, cplx* __restrict__ val_all) {
    const int k = members.k[blockIdx.x];
    const cplx* __restrict__ row = rows + (int64_t)k * M;
    const double* __restrict__ th = th_red + (int64_t)k * M;
    cplx* __restrict__ recov = recov_all + (int64_t)blockIdx.x * g.ntau * g.nfd;
    int64_t* __restrict__ pix = pix_all + (int64_t)blockIdx.x * M;
    cplx* __restrict__ val = val_all + (int64_t)blockIdx.x * M;
    const int i = N / 2;
    const double ti = th[i], two_eta = 2 * eta;
    for (int j = threadIdx.x; j < N; j += 256) {
        int64_t o = -1;
        cplx v = mk(0.0, 0.0);
        if (j != i) {
            const double tj = th[j];
            const int64_t bx = hist_bin(tj - ti, g.fd0, g.fd1_step, g.nfd);
            const int64_t by = hist_bin(eta * (tj * tj - ti * ti), g.tau0, g.tau1_step, g.ntau);
            if (bx >= 0 && by >= 0) {
                o = by * g.nfd + bx;
                const double scl = rsqrt(fabs(two_eta * (ti - tj)));      // thth / sqrt(|2 eta fd_map.T|)  (ththmod.py:226), as rev_gather_body
                v = mk(row[j].x * scl, row[j].y * scl);
            }
        }
        pix[j] = o; val[j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256) {
        const int64_t o = pix[j];
        if (o < 0 || o == centre) continue;
        bool head = true;
        for (int r = 1; r <= kRowRun && j - r >= 0; ++r) head = head && pix[j - r] != o;
        if (!head) continue;
        double sr = val[j].x, si = val[j].y;
        for (int r = 1; r <= kRowRun && j + r < N; ++r)
            if (pix[j + r] == o) { sr = sr + val[j + r].x; si = si + val[j + r].y; }
        const double scl = 1.0 / (double)cnt[o];                      // NumPy divides complex by real as v * (1 / c)
        auto clean = [](double x) { return x != x ? 0.0 : (x == INFINITY ? 1.7976931348623157e308 : (x == -INFINITY ? -1.7976931348623157e308 : x)); };
        recov[o] = mk(clean(sr * scl), clean(si * scl));
    }
}
}  // namespace scint

extern "C" int32_t scint_retrieval_tail_workspace_bytes(int64_t M, int64_t ntau, int64_t nfd, size_t* bytes) {
    SCINT_REQUIRE(bytes && M >= 1 && ntau >= 1 && nfd >= 1, "retrieval_tail_workspace_bytes: bad arguments");
    size_t fft = 0;
    const int32_t rc = scint_fft2_workspace_bytes(ntau, nfd, &fft);
    if (rc != SCINT_OK) return rc;
    *bytes = align_up(sizeof(uint32_t) * (size_t)ntau * (size_t)nfd, 256) + align_up(sizeof(cplx) * (size_t)ntau * (size_t)nfd * kTailGroup, 256) +
             align_up(fft, 256) + align_up((sizeof(int64_t) + sizeof(cplx)) * (size_t)M * kTailGroup, 256) + 512;
    return SCINT_OK;
}

// Per chunk k with keep_n[k] >= 2 (ththmod.py:1457-1470): theta-theta of the E field = zeros with row N/2 = rows[k][:N] (the caller's
// conj(V) sqrt(w)), its non-Hermitian back-map on the chunk's axes (one-row form above), scale * ifft2(ifftshift(.))[:nf, :nt] -> out[k].
// class_id (HOST [n]): chunks with equal ids in a row share theta grid, curvature and axes (the caller's promise: the pair counts
// are formed from the first one's).  Chunks with keep_n[k] < 2 (failed or skipped by the caller) are left as the caller initialised them.
extern "C" int32_t scint_retrieval_tail(const scint_c128* rows, const double* th_red, const int32_t* keep_n, const int32_t* class_id,
                                        const scint_cs_geom* geoms, const double* etas, int64_t n, int64_t M, int64_t nf, int64_t nt,
                                        double scale, scint_c128* out, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SCINT_REQUIRE(rows && th_red && keep_n && class_id && geoms && etas && out && workspace && n >= 1 && M >= 1, "retrieval_tail: null pointer or empty batch");
    const int64_t ntau = geoms[0].ntau, nfd = geoms[0].nfd, npix = ntau * nfd;
    size_t need = 0, fft = 0;
    if (scint_retrieval_tail_workspace_bytes(M, ntau, nfd, &need) != SCINT_OK || scint_fft2_workspace_bytes(ntau, nfd, &fft) != SCINT_OK) return SCINT_E_ARG;
    if (workspace_bytes < need) { set_error("scint: retrieval_tail workspace too small"); return SCINT_E_WORKSPACE; }
    char* base = (char*)workspace;
    uint32_t* cnt = (uint32_t*)base;
    cplx* recov = (cplx*)(base + align_up(sizeof(uint32_t) * (size_t)npix, 256));
    char* fftws = (char*)recov + align_up(sizeof(cplx) * (size_t)npix * kTailGroup, 256);
    int64_t* pix = (int64_t*)(fftws + align_up(fft, 256));
    cplx* val = (cplx*)(pix + (size_t)M * kTailGroup);
    int64_t k = 0;
    while (k < n) {
        int64_t e = k;
        while (e < n && class_id[e] == class_id[k]) ++e;                  // the class: chunks k .. e-1
        int64_t first = -1;
        for (int64_t m = k; m < e; ++m) if (keep_n[m] >= 2) { first = m; break; }
        if (first >= 0) {
            const int64_t N = keep_n[first];
            SCINT_REQUIRE(N <= M && geoms[first].ntau == ntau && geoms[first].nfd == nfd, "retrieval_tail: chunk does not match the batch's shape");
            const GeomDev g = to_dev(geoms[first]);
            const double eta = etas[first];
            const int64_t cbx = hist_bin(0.0, g.fd0, g.fd1_step, g.nfd), cby = hist_bin(eta * 0.0, g.tau0, g.tau1_step, g.ntau);
            const int64_t centre = (cbx >= 0 && cby >= 0) ? cby * g.nfd + cbx : -1;
            SCINT_HIP(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (size_t)npix, stream));
            const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(N * N, 256 * 4), 4096);
            hipLaunchKernelGGL(rev_count_kernel, dim3(blocks), dim3(256), 0, stream, th_red + first * M, (int)N, eta, g, cnt);
            SCINT_LAUNCH_CHECK();
            for (int64_t m0 = k; m0 < e; m0 += kTailGroup) {
                TailMembers members;
                int cntm = 0;
                for (int64_t m = m0; m < std::min(e, m0 + kTailGroup); ++m)
                    if (keep_n[m] >= 2) {
                        SCINT_REQUIRE(keep_n[m] == N, "retrieval_tail: chunks of one class keep different numbers of centres");
                        members.k[cntm++] = (int32_t)m;
                    }
                if (cntm == 0) continue;
                SCINT_HIP(hipMemsetAsync(recov, 0, sizeof(cplx) * (size_t)npix * (size_t)cntm, stream));
                hipLaunchKernelGGL(rev_row_kernel, dim3((unsigned)cntm), dim3(256), 0, stream, (const cplx*)rows, th_red, M, members, (int)N, eta, g,
                                   cnt, recov, centre, pix, val);
                SCINT_LAUNCH_CHECK();
                for (int q = 0; q < cntm; ++q) {
                    const int32_t rc = scint_ifft2_shifted((const scint_c128*)(recov + (size_t)q * (size_t)npix), ntau, nfd, scale, nf, nt,
                                                           out + (int64_t)members.k[q] * nf * nt, fftws, fft, stream_);
                    if (rc != SCINT_OK) return rc;
                }
            }
        }
        k = e;
    }
    return SCINT_OK;
}
