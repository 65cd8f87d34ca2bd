"""Scratch experiment (host interpreter, no GPU): passes per curvature of the eigenPAIR sweep and the deviation of the rank-1
model |w| V V^H from the oracle's (ARPACK) for the build constant SCINT_VEC_GAP_FACTOR.

    SCINT_EMU_DEFINES=-DSCINT_VEC_GAP_FACTOR=100.0 python tools/experiments/vec_factor_passes.py [size] [kind]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "emu"))


class MP:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    kind = sys.argv[2] if len(sys.argv) > 2 else "arc"
    import emulated
    emulated.install(MP())
    from oracle import thth_oracle as to
    from scintools_amd import ththmod
    if kind == "arc":
        from scintools_amd.synth import arc_dynspec
        dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
    else:
        from oracle import sim_oracle
        sim = sim_oracle.baseline_dynspec(size, 3)
        dyn, freqs, times, eta_true = np.array(sim.dyn, dtype=float), sim.freqs, sim.times, float(sim.eta)
    dyn = dyn - dyn.mean()
    fd = to.fft_axis(times, 1000.0, 0)
    tau = to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    etas = np.geomspace(0.25, 4.0, 9) * eta_true
    CS = to.conjugate_spectrum(dyn, 0)
    w, V_t, info = ththmod.eigvec_sweep(CS, tau, fd, etas, edges)
    V = V_t.numpy()
    devs = []
    for i, eta in enumerate(etas):
        red, _ = to.thth_redmap(CS, tau, fd, eta, edges)
        n = red.shape[0]
        ev, U = np.linalg.eigh(red)
        u = U[:, -1]
        v = V[i, :n]
        ph = np.vdot(u, v)
        ph /= abs(ph)
        m_ref = np.outer(u, u.conj())
        m_got = np.outer(v / ph, (v / ph).conj())
        devs.append((abs(w[i] - ev[-1]) / ev[-1], np.abs(m_got - m_ref).max() / np.abs(m_ref).max(),
                     (ev[-1] - ev[-2]) / ev[-1]))
    print("defines", os.environ.get("SCINT_EMU_DEFINES", ""), "size", size, kind)
    print("iters", info["iters"].tolist(), "mean", info["iters"].mean())
    for i, d in enumerate(devs):
        print(f"  eta/eta_true {etas[i]/eta_true:5.2f} N {info['N'][i]:5d} iters {info['iters'][i]:3d} "
              f"dw {d[0]:.1e} dmodel/max {d[1]:.1e} relgap {d[2]:.3f}")


if __name__ == "__main__":
    main()
