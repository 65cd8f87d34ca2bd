// thth.hpp -- CS <-> theta-theta maps (ththmod.py:56-271) as gfx950 kernels.
#pragma once
#include "common.hpp"

namespace scint {

// Device-side copy of the CS geometry plus the derived constants the reference
// computes once per call (ththmod.py:90-91): all host-computed with NumPy so the
// kernels see bit-identical operands.
struct GeomDev {
    int64_t ntau, nfd;
    double tau0, dtau, half_dtau;  // tau[0], diff(tau).mean(), dtau/2
    double fd0, dfd, half_dfd;     // fd[0],  diff(fd).mean(),  dfd/2
    double tau1_step, fd1_step;    // tau[1]-tau[0], fd[1]-fd[0]   (rev_map)
};
inline GeomDev to_dev(const scint_cs_geom& g) {
    GeomDev d;
    d.ntau = g.ntau; d.nfd = g.nfd;
    d.tau0 = g.tau0; d.dtau = g.dtau; d.half_dtau = g.dtau / 2;
    d.fd0 = g.fd0; d.dfd = g.dfd; d.half_dfd = g.dfd / 2;
    d.tau1_step = g.tau1_step; d.fd1_step = g.fd1_step;
    return d;
}

// One theta-theta matrix to build: curvature, crop and destination.
struct GatherJob {
    double eta, two_eta;     // eta and 2*eta (ththmod.py:95, 107)
    const int32_t* keep;     // N indices into th_cents (ascending)
    int32_t n;               // N
    int32_t hermitian;
    cplx* out;               // [N, ld]
    int64_t ld;
};

// Enqueue the gather for `njobs` jobs (device array) on `stream`; `nmax` = max N.
int32_t launch_gather(const cplx* cs, const GeomDev& g, const double* th_cents, int64_t M,
                      const GatherJob* jobs_dev, int njobs, int nmax, hipStream_t stream);

}  // namespace scint
