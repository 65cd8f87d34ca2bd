// arcnorm.hip -- arc normalisation of the secondary spectrum (gfx950).
//
//   scint_spline_resample  Dynspec.scale_dyn(scale='lambda')   dynspec.py:3948-3957
//   scint_norm_sspec       the row loop of Dynspec.norm_sspec   dynspec.py:2093-2127
//   scint_masked_colavg    np.ma.average(..., axis=0, weights)  dynspec.py:2171-2181
//   scint_row_nanmean      delay response of subtract_artefacts dynspec.py:2060-2061
//   scint_block_std        fit_arc's noise estimate             dynspec.py:1097-1101
//
// Everything here is HBM-bound streaming work (a gather along each delay row, column and row
// reductions).  Compiled with -ffp-contract=off: the interpolation must round like NumPy's
// arr_interp (slope*(x - xp[j]) + fp[j], no fused multiply-add).
#include <math.h>

#include <algorithm>

#include "common.hpp"

namespace scint {

// ------------------------------------------------------------------------------
// cubic-spline resample down the frequency axis
// ------------------------------------------------------------------------------
struct SplineSys {
    const double* h;    // [nf-1] knot spacings
    const double* sub;  // [nf] sub-diagonal a_i of the interior rows 1..nf-2
    const double* inv;  // [nf] 1 / pivot
    const double* sup;  // [nf] Thomas c'_i
    double e0, e1, e2, e3;
};

// Thomas sweeps of the moment system, one thread per (time column, frequency block); loads
// coalesce across the wavefront along time.  Both recurrences contract (|sub*inv|, |sup| are
// about 0.27 on a uniform axis), so a block does not wait for its neighbour: it starts `warm`
// rows early from zero and the start-up error has decayed below rounding (the host sizes `warm`
// from the actual factors) by the first row it stores.  blockIdx.y = frequency block.
__global__ void __launch_bounds__(64)
spline_forward_kernel(const double* dyn, int64_t nf, int64_t nt, int reverse, SplineSys s,
                      int64_t block_rows, int64_t warm, double* D) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= nt) return;
    const int64_t first = 1 + (int64_t)blockIdx.y * block_rows;       // rows [first, last) are stored
    const int64_t last = min(first + block_rows, nf - 1);
    const int64_t i0 = max((int64_t)1, first - warm);
    auto row = [&](int64_t i) { return (reverse ? nf - 1 - i : i) * nt + t; };
    double y_prev = gload(dyn + row(i0 - 1)), y_cur = gload(dyn + row(i0));
    double dp = 0.0;
    for (int64_t i = i0; i < last; ++i) {
        const double y_next = gload(dyn + row(i + 1));
        const double r = 6.0 * ((y_next - y_cur) / gload(s.h + i) - (y_cur - y_prev) / gload(s.h + i - 1));
        dp = (r - gload(s.sub + i) * dp) * gload(s.inv + i);
        if (i >= first) gstore(D + i * nt + t, dp);
        y_prev = y_cur; y_cur = y_next;
    }
}

__global__ void __launch_bounds__(64)
spline_backward_kernel(const double* D, int64_t nf, int64_t nt, SplineSys s, int64_t block_rows,
                       int64_t warm, double* M) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= nt) return;
    const int64_t first = 1 + (int64_t)blockIdx.y * block_rows;
    const int64_t last = min(first + block_rows, nf - 1);
    const int64_t i1 = min(nf - 2, last - 1 + warm);
    double m_next = 0.0;   // exact at i1 == nf-2 (sup[nf-2] == 0), decayed away otherwise
    for (int64_t i = i1; i >= first; --i) {
        m_next = gload(D + i * nt + t) - gload(s.sup + i) * m_next;
        if (i < last) gstore(M + i * nt + t, m_next);
    }
}

// not-a-knot ends: M[0] and M[nf-1] follow from their two neighbours
__global__ void __launch_bounds__(256) spline_ends_kernel(int64_t nf, int64_t nt, SplineSys s, double* M) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    gstore(M + t, s.e0 * gload(M + nt + t) + s.e1 * gload(M + 2 * nt + t));
    gstore(M + (nf - 1) * nt + t, s.e2 * gload(M + (nf - 2) * nt + t) + s.e3 * gload(M + (nf - 3) * nt + t));
}

__global__ void __launch_bounds__(256)
spline_eval_kernel(const double* dyn, int64_t nf, int64_t nt, int reverse, const double* M,
                   const int32_t* idx, const double* coef, int64_t nout, double* out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k = blockIdx.y;
    if (t >= nt) return;
    const int64_t i = idx[k];
    const int64_t r0 = reverse ? nf - 1 - i : i, r1 = reverse ? nf - 2 - i : i + 1;
    const double c0 = coef[4 * k], c1 = coef[4 * k + 1], c2 = coef[4 * k + 2], c3 = coef[4 * k + 3];
    const double v = c0 * gload(dyn + r0 * nt + t) + c1 * gload(dyn + r1 * nt + t) +
                     c2 * gload(M + i * nt + t) + c3 * gload(M + (i + 1) * nt + t);
    gstore(out + (nout - 1 - k) * nt + t, v);  // np.flipud
}

// ------------------------------------------------------------------------------
// norm_sspec rows
// ------------------------------------------------------------------------------
struct NormRow {
    const double* row;     // sspec row
    const double* fdop;
    double scale, offset;
    int64_t c0, c1;        // selected columns [c0, c1]
    int64_t cut_lo, cut_hi;
    bool has_offset;
    __device__ inline double xp(int64_t c) const { return gload(fdop + c) / scale; }
    __device__ inline double fp(int64_t c) const {
        if (c >= cut_lo && c < cut_hi) return NAN;
        const double v = gload(row + c);
        return has_offset ? v - offset : v;
    }
    // np.interp(x, xp, fp) for one x (numpy/_core/src/multiarray/compiled_base.c, arr_interp)
    __device__ inline double interp(double x, double x_lo, double x_hi, double inv_step) const {
        const int64_t n = c1 - c0 + 1;
        if (n == 1) return fp(c0);
        if (x != x) return x;
        if (x > x_hi) return fp(c1);
        if (x < x_lo) return fp(c0);
        // j = largest column with xp(j) <= x: uniform-axis guess, short walk, bisection fallback.
        // xp(j) and xp(j+1) are carried along so that the usual case costs two divisions.
        const double g = floor((x - x_lo) * inv_step);
        int64_t j = c0 + (int64_t)fmin(fmax(g, 0.0), (double)(n - 1));
        double xj = xp(j), xj1 = 0.0;
        bool have1 = false;
        int walk = 0;
        while (xj > x && j > c0 && walk < 6) { xj1 = xj; have1 = true; --j; xj = xp(j); ++walk; }
        if (!(xj > x)) {
            while (j < c1 && walk < 6) {
                if (!have1) { xj1 = xp(j + 1); have1 = true; }
                if (xj1 > x) break;
                ++j; xj = xj1; have1 = false; ++walk;
            }
        }
        if (xj > x || (j < c1 && (have1 ? xj1 : xp(j + 1)) <= x)) {
            int64_t lo = c0, hi = c1 + 1;  // first column with xp > x is in (lo, hi]
            while (lo < hi) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if (x >= xp(mid)) lo = mid + 1; else hi = mid;
            }
            j = lo - 1;
            xj = xp(j);
            have1 = false;
        }
        if (j == c1) return fp(j);
        const double fj = fp(j);
        if (xj == x) return fj;
        if (!have1) xj1 = xp(j + 1);
        const double fj1 = fp(j + 1);
        const double slope = (fj1 - fj) / (xj1 - xj);
        double r = slope * (x - xj) + fj;
        if (r != r) {
            r = slope * (x - xj1) + fj1;
            if (r != r && fj == fj1) r = fj;
        }
        return r;
    }
};

struct NormParams {
    const double* sspec; int64_t ld, nc;
    const double* fdop; const double* yaxis;
    int64_t row0, nr;
    double eta, maxnormfac;
    int64_t cut_lo, cut_hi;
    const double* row_offset;
    const double* x; const double* xlin; int64_t nx;
    double* norm; uint8_t* mask; double* pow;
};

// one workgroup per delay row
__global__ void __launch_bounds__(256) norm_sspec_kernel(NormParams p) {
    __shared__ double red[4];
    const int64_t r = blockIdx.x;
    const double scale = sqrt(gload(p.yaxis + p.row0 + r) / p.eta);
    const double lim = p.maxnormfac * scale;
    // sel = |fdop| <= lim on an ascending axis: first column >= -lim .. last column <= lim
    int64_t lo = 0, hi = p.nc;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (gload(p.fdop + m) >= -lim) hi = m; else lo = m + 1; }
    const int64_t c0 = lo;
    lo = 0; hi = p.nc;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (gload(p.fdop + m) <= lim) lo = m + 1; else hi = m; }
    const int64_t c1 = lo - 1;
    NormRow row;
    row.row = p.sspec + (p.row0 + r) * p.ld;
    row.fdop = p.fdop; row.scale = scale;
    row.has_offset = p.row_offset != nullptr;
    row.offset = row.has_offset ? gload(p.row_offset + r) : 0.0;
    row.c0 = c0; row.c1 = c1; row.cut_lo = p.cut_lo; row.cut_hi = p.cut_hi;
    const bool empty = c1 < c0 || !(lim == lim);
    double x_lo = 0.0, x_hi = 0.0, inv_step = 0.0, x_abs = 0.0;
    if (!empty) {
        x_lo = row.xp(c0); x_hi = row.xp(c1);
        inv_step = (c1 > c0 && x_hi > x_lo) ? (double)(c1 - c0) / (x_hi - x_lo) : 0.0;
        x_abs = fmax(fabs(x_lo), fabs(x_hi));
    }
    double acc = 0.0, cnt = 0.0;
    for (int64_t k = threadIdx.x; k < p.nx; k += 256) {
        const double x = gload(p.x + k);
        double v = NAN, vp = NAN;
        bool m = true;
        if (!empty) {
            v = row.interp(x, x_lo, x_hi, inv_step);
            m = (fabs(x) > x_abs) || (v != v);
            vp = p.xlin ? row.interp(gload(p.xlin + k), x_lo, x_hi, inv_step) : v;
        }
        p.norm[r * p.nx + k] = v;
        p.mask[r * p.nx + k] = m ? 1 : 0;
        // the masked division normSspec/10 masks every non-finite entry (dynspec.py:2118-2127)
        if (!m && isfinite(vp)) { acc += pow(10.0, vp / 10.0); cnt += 1.0; }
    }
    acc = block_sum(acc, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) p.pow[r] = cnt > 0.0 ? acc / cnt : NAN;
}

// ------------------------------------------------------------------------------
// masked weighted column average
// ------------------------------------------------------------------------------
constexpr int kAvgRows = 32;  // rows per partial

__global__ void __launch_bounds__(256)
colavg_partial_kernel(const double* norm, const uint8_t* mask, int64_t nr, int64_t nx, const double* w,
                      const uint8_t* rowsel, double* part /*[nchunk][3][nx]*/) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nx) return;
    const int64_t r0 = (int64_t)blockIdx.y * kAvgRows, r1 = min(r0 + kAvgRows, nr);
    double num = 0.0, den = 0.0, cnt = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        if (rowsel && !rowsel[r]) continue;
        if (mask[r * nx + k]) continue;
        const double wr = w[r];
        num += norm[r * nx + k] * wr;
        den += wr;
        cnt += 1.0;
    }
    part[((int64_t)blockIdx.y * 3) * nx + k] = num;
    part[((int64_t)blockIdx.y * 3 + 1) * nx + k] = den;
    part[((int64_t)blockIdx.y * 3 + 2) * nx + k] = cnt;
}

__global__ void __launch_bounds__(256)
colavg_final_kernel(const double* part, int64_t nchunk, int64_t nx, double* avg, uint8_t* empty) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nx) return;
    double num = 0.0, den = 0.0, cnt = 0.0;
    for (int64_t c = 0; c < nchunk; ++c) {
        num += part[(c * 3) * nx + k];
        den += part[(c * 3 + 1) * nx + k];
        cnt += part[(c * 3 + 2) * nx + k];
    }
    // a column with no entry is masked in the reference, with 0.0 left under the mask
    avg[k] = cnt > 0.0 ? num / den : 0.0;
    empty[k] = cnt > 0.0 ? 0 : 1;
}

// ------------------------------------------------------------------------------
// row nanmean, block std
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
row_nanmean_kernel(const double* sspec, int64_t ld, int64_t nc, int64_t row0, const uint8_t* colsel,
                   int64_t cut_lo, int64_t cut_hi, double* out) {
    __shared__ double red[4];
    const double* row = sspec + (row0 + blockIdx.x) * ld;
    double acc = 0.0, cnt = 0.0;
    for (int64_t c = threadIdx.x; c < nc; c += 256) {
        if (!colsel[c] || (c >= cut_lo && c < cut_hi)) continue;
        const double v = row[c];
        if (v == v) { acc += v; cnt += 1.0; }
    }
    acc = block_sum(acc, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) out[blockIdx.x] = acc / cnt;
}

constexpr int kStdBlocks = 1024;

// pass 0: sum(x); pass 1: sum((x - mean)^2), mean = *mean_p
__global__ void __launch_bounds__(256)
block_moment_kernel(const double* a, int64_t ld, int64_t r0, int64_t rows, int64_t c_lo, int64_t c_hi,
                    int64_t nc, const double* mean_p, double* partial) {
    __shared__ double red[4];
    const int64_t width = c_lo + (nc - c_hi);
    const int64_t total = rows * width;
    const double mean = mean_p ? mean_p[0] : 0.0;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / width, q = i - r * width;
        const int64_t c = q < c_lo ? q : q - c_lo + c_hi;
        const double v = a[(r0 + r) * ld + c];
        acc += mean_p ? (v - mean) * (v - mean) : v;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256)
block_moment_final_kernel(const double* partial, int np, double inv_n, int take_sqrt, double* out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = take_sqrt ? sqrt(acc * inv_n) : acc * inv_n;
}

}  // namespace scint

using namespace scint;

extern "C" int32_t scint_spline_resample(const double* dyn, int64_t nf, int64_t nt, int32_t reverse,
                                         const double* h, const double* sub, const double* inv,
                                         const double* sup, const double* end, int64_t block_rows,
                                         int64_t warm, const int32_t* idx, const double* coef,
                                         int64_t nout, double* out, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(dyn && h && sub && inv && sup && end && idx && coef && out && workspace,
                  "spline_resample: null pointer");
    SCINT_REQUIRE(nf >= 4 && nt >= 1 && nout >= 1, "spline_resample: need at least 4 channels");
    SCINT_REQUIRE(nout <= 65535, "spline_resample: too many output rows");
    SCINT_REQUIRE(warm >= 0, "spline_resample: bad warm-up length");
    if (workspace_bytes < 2 * sizeof(double) * (size_t)nf * (size_t)nt) {
        set_error("scint: spline_resample workspace too small");
        return SCINT_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    SplineSys s{h, sub, inv, sup, end[0], end[1], end[2], end[3]};
    double* D = (double*)workspace;
    double* M = D + nf * nt;
    const int64_t rows = nf - 2;                       // interior rows 1..nf-2
    if (block_rows <= 0 || block_rows > rows) block_rows = rows;   // one block = the plain sweep
    const int64_t nblk = ceil_div(rows, block_rows);
    SCINT_REQUIRE(nblk <= 65535, "spline_resample: too many frequency blocks");
    const dim3 grid((unsigned)ceil_div(nt, 64), (unsigned)nblk);
    hipLaunchKernelGGL(spline_forward_kernel, grid, dim3(64), 0, stream, dyn, nf, nt, (int)reverse, s,
                       block_rows, warm, D);
    SCINT_LAUNCH_CHECK();
    hipLaunchKernelGGL(spline_backward_kernel, grid, dim3(64), 0, stream, D, nf, nt, s, block_rows, warm, M);
    SCINT_LAUNCH_CHECK();
    hipLaunchKernelGGL(spline_ends_kernel, dim3((unsigned)ceil_div(nt, 256)), dim3(256), 0, stream, nf, nt, s, M);
    SCINT_LAUNCH_CHECK();
    hipLaunchKernelGGL(spline_eval_kernel, dim3((unsigned)ceil_div(nt, 256), (unsigned)nout), dim3(256), 0,
                       stream, dyn, nf, nt, (int)reverse, M, idx, coef, nout, out);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

extern "C" int32_t scint_norm_sspec(const double* sspec, int64_t ld, int64_t nc, const double* fdop,
                                    const double* yaxis, int64_t row0, int64_t nr, double eta,
                                    double maxnormfac, int64_t cut_lo, int64_t cut_hi,
                                    const double* row_offset, const double* x, const double* xlin,
                                    int64_t nx, double* norm_out, uint8_t* mask_out, double* pow_out,
                                    void* stream_) {
    SCINT_REQUIRE(sspec && fdop && yaxis && x && norm_out && mask_out && pow_out, "norm_sspec: null pointer");
    SCINT_REQUIRE(nc >= 1 && ld >= nc && row0 >= 0 && nr >= 0 && nx >= 1, "norm_sspec: bad sizes");
    if (nr == 0) return SCINT_OK;
    NormParams p;
    p.sspec = sspec; p.ld = ld; p.nc = nc; p.fdop = fdop; p.yaxis = yaxis;
    p.row0 = row0; p.nr = nr; p.eta = eta; p.maxnormfac = maxnormfac;
    p.cut_lo = cut_lo; p.cut_hi = cut_hi; p.row_offset = row_offset;
    p.x = x; p.xlin = xlin; p.nx = nx;
    p.norm = norm_out; p.mask = mask_out; p.pow = pow_out;
    hipLaunchKernelGGL(norm_sspec_kernel, dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream_, p);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

extern "C" int32_t scint_masked_colavg_workspace_bytes(int64_t nr, int64_t nx, size_t* bytes) {
    SCINT_REQUIRE(bytes && nr >= 0 && nx >= 1, "masked_colavg_workspace_bytes: bad arguments");
    *bytes = sizeof(double) * 3 * (size_t)std::max<int64_t>(1, ceil_div(nr, kAvgRows)) * (size_t)nx;
    return SCINT_OK;
}

extern "C" int32_t scint_masked_colavg(const double* norm, const uint8_t* mask, int64_t nr, int64_t nx,
                                       const double* weights, const uint8_t* rowsel, double* avg_out,
                                       uint8_t* empty_out, void* workspace, size_t workspace_bytes,
                                       void* stream_) {
    SCINT_REQUIRE(norm && mask && weights && avg_out && empty_out && workspace, "masked_colavg: null pointer");
    SCINT_REQUIRE(nr >= 1 && nx >= 1, "masked_colavg: bad sizes");
    size_t need = 0;
    scint_masked_colavg_workspace_bytes(nr, nx, &need);
    if (workspace_bytes < need) { set_error("scint: masked_colavg workspace too small"); return SCINT_E_WORKSPACE; }
    const int64_t nchunk = ceil_div(nr, kAvgRows);
    SCINT_REQUIRE(nchunk <= 65535, "masked_colavg: too many rows");
    hipStream_t stream = (hipStream_t)stream_;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(colavg_partial_kernel, dim3((unsigned)ceil_div(nx, 256), (unsigned)nchunk), dim3(256), 0,
                       stream, norm, mask, nr, nx, weights, rowsel, part);
    SCINT_LAUNCH_CHECK();
    hipLaunchKernelGGL(colavg_final_kernel, dim3((unsigned)ceil_div(nx, 256)), dim3(256), 0, stream, part,
                       nchunk, nx, avg_out, empty_out);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

extern "C" int32_t scint_row_nanmean(const double* sspec, int64_t ld, int64_t nc, int64_t row0, int64_t nr,
                                     const uint8_t* colsel, int64_t cut_lo, int64_t cut_hi, double* out,
                                     void* stream_) {
    SCINT_REQUIRE(sspec && colsel && out, "row_nanmean: null pointer");
    SCINT_REQUIRE(nc >= 1 && ld >= nc && row0 >= 0 && nr >= 0, "row_nanmean: bad sizes");
    if (nr == 0) return SCINT_OK;
    hipLaunchKernelGGL(row_nanmean_kernel, dim3((unsigned)nr), dim3(256), 0, (hipStream_t)stream_, sspec, ld, nc,
                       row0, colsel, cut_lo, cut_hi, out);
    SCINT_LAUNCH_CHECK();
    return SCINT_OK;
}

extern "C" int32_t scint_block_std(const double* a, int64_t ld, int64_t nc, int64_t r0, int64_t r1,
                                   int64_t c_lo, int64_t c_hi, double* out, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    SCINT_REQUIRE(a && out && workspace, "block_std: null pointer");
    SCINT_REQUIRE(nc >= 1 && ld >= nc && r0 >= 0 && r1 > r0 && c_lo >= 0 && c_hi >= c_lo && c_hi <= nc,
                  "block_std: bad ranges");
    const int64_t total = (r1 - r0) * (c_lo + nc - c_hi);
    SCINT_REQUIRE(total >= 1, "block_std: empty selection");
    if (workspace_bytes < sizeof(double) * (kStdBlocks + 8)) {
        set_error("scint: block_std workspace too small");
        return SCINT_E_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    double* partial = (double*)workspace;
    double* mean = partial + kStdBlocks;
    const int blocks = (int)std::min<int64_t>(kStdBlocks, std::max<int64_t>(1, ceil_div(total, 1024)));
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(block_moment_kernel, dim3(blocks), dim3(256), 0, stream, a, ld, r0, r1 - r0, c_lo, c_hi,
                           nc, pass ? mean : (const double*)nullptr, partial);
        SCINT_LAUNCH_CHECK();
        hipLaunchKernelGGL(block_moment_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks,
                           1.0 / (double)total, pass, pass ? out : mean);
        SCINT_LAUNCH_CHECK();
    }
    return SCINT_OK;
}
