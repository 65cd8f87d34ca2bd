// Host-side entry points around the W-vector block algebra of scintools_amd/csrc/blockw.hpp, so that
// tests/test_blockw_cpu.py can compare it with tools/models/blockw_reference.py without a GPU.
#include "../../scintools_amd/csrc/blockw.hpp"

using namespace scint;

template <int W>
static void from_sums(const double* sa, const double* sg, double* a_out, double* b_out, double* inv_out) {
    const BlkW<W> k = bw_from_sums<W>(sa, sg);
    for (int r = 0; r < W; ++r) {
        inv_out[r] = k.inv[r];
        for (int c = 0; c < W; ++c) {
            const cplx a = bw_a<W>(k, r, c);
            a_out[2 * (r * W + c)] = a.x; a_out[2 * (r * W + c) + 1] = a.y;
            const cplx b = r <= c ? k.b[r][c] : mk(0.0, 0.0);
            b_out[2 * (r * W + c)] = b.x; b_out[2 * (r * W + c) + 1] = b.y;
        }
    }
}
template <int W>
static void q_row(const double* sa, const double* sg, const double* u, const double* q, double* x_out, double* qbh_out) {
    const BlkW<W> k = bw_from_sums<W>(sa, sg);
    cplx uu[W], qq[W], xx[W], hh[W];
    for (int c = 0; c < W; ++c) { uu[c] = mk(u[2 * c], u[2 * c + 1]); qq[c] = mk(q[2 * c], q[2 * c + 1]); }
    bw_q_row<W>(k, uu, qq, xx);
    bw_qbh_row<W>(k, qq, hh);
    for (int c = 0; c < W; ++c) { x_out[2 * c] = xx[c].x; x_out[2 * c + 1] = xx[c].y; qbh_out[2 * c] = hh[c].x; qbh_out[2 * c + 1] = hh[c].y; }
}
template <int W>
static void band(const double* pa, const double* pb, int nblk, double* band_out) {
    const int n = W * nblk;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k <= W; ++k) {
            const int j = i / W, r = i - j * W;
            const cplx v = bw_band_entry<W>(pa + (size_t)j * W * W, j + 1 < nblk ? pb + (size_t)j * W * W : nullptr, r, k);
            band_out[2 * (i * (W + 1) + k)] = v.x; band_out[2 * (i * (W + 1) + k) + 1] = v.y;
        }
}
template <int W>
static int count(const double* band_in, int n, double x, double tiny) {
    return bw_band_count<W>((const cplx*)band_in, n, x, tiny);
}
template <int W>
static double invit(const double* band_in, int n, double sigma, double tiny, double* s_out, double* work) {
    double* d = work;
    cplx* m = (cplx*)(work + n + (n & 1));
    bw_band_count<W>((const cplx*)band_in, n, sigma, tiny, d, m);
    return bw_inverse_iteration<W>(d, m, n, (cplx*)s_out);
}

#define DISPATCH(W, call2, call3, call4) do { if ((W) == 2) { call2; } else if ((W) == 3) { call3; } else { call4; } } while (0)

extern "C" {
void bw_from_sums_c(int W, const double* sa, const double* sg, double* a, double* b, double* inv) {
    DISPATCH(W, from_sums<2>(sa, sg, a, b, inv), from_sums<3>(sa, sg, a, b, inv), from_sums<4>(sa, sg, a, b, inv));
}
void bw_q_row_c(int W, const double* sa, const double* sg, const double* u, const double* q, double* x, double* qbh) {
    DISPATCH(W, q_row<2>(sa, sg, u, q, x, qbh), q_row<3>(sa, sg, u, q, x, qbh), q_row<4>(sa, sg, u, q, x, qbh));
}
void bw_band_c(int W, const double* pa, const double* pb, int nblk, double* out) {
    DISPATCH(W, band<2>(pa, pb, nblk, out), band<3>(pa, pb, nblk, out), band<4>(pa, pb, nblk, out));
}
int bw_count_c(int W, const double* band_in, int n, double x, double tiny) {
    int r = 0;
    DISPATCH(W, r = count<2>(band_in, n, x, tiny), r = count<3>(band_in, n, x, tiny), r = count<4>(band_in, n, x, tiny));
    return r;
}
double bw_invit_c(int W, const double* band_in, int n, double sigma, double tiny, double* s, double* work) {
    double r = 0;
    DISPATCH(W, r = invit<2>(band_in, n, sigma, tiny, s, work), r = invit<3>(band_in, n, sigma, tiny, s, work),
             r = invit<4>(band_in, n, sigma, tiny, s, work));
    return r;
}
}
