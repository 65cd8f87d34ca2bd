"""TEST INFRASTRUCTURE ONLY: run a fixed set of hot-path calls on the host interpreter and save
every result (argv[1] = output .npz).  tests/test_emu_cpu.py runs it twice -- SCINT_EMU_ORDER unset
and =rev (waves, lanes and blocks scheduled in the opposite order) -- and demands identical bits:
every order is a legal GPU schedule, so a difference would mean a missing barrier, an inter-block
dependence or reliance on lock-step lanes."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def main(out_path):
    import torch
    from _pytest.monkeypatch import MonkeyPatch
    import emulated
    patch = MonkeyPatch()
    emulated.install(patch)
    from oracle import thth_oracle
    from scintools_amd import ththmod
    from scintools_amd.dynspec import sspec_device
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(96, 80, seed=5, nimg=12)
    dyn = dyn - dyn.mean()
    fd = thth_oracle.fft_axis(times, 1000.0, 1)
    tau = thth_oracle.fft_axis(freqs, 1.0, 1)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 100)
    etas = np.geomspace(0.5, 2.0, 5) * eta_true
    res = {}
    cs_t = ththmod.conjugate_spectrum(dyn, 1, pad_value=0.0)
    CS = cs_t.cpu().numpy()
    res["cs"] = CS
    res["sspec"] = sspec_device(ththmod.to_device(dyn, torch.float64)).cpu().numpy()
    res["sspec_pw"] = sspec_device(ththmod.to_device(dyn, torch.float64), prewhite=True).cpu().numpy()
    res["red"] = ththmod.thth_redmap(CS, tau, fd, etas[2], edges)[0]
    res["red_nh"] = ththmod.thth_redmap(CS, tau, fd, etas[2], edges, False)[0]
    eigs, info = ththmod.eval_sweep(cs_t, tau, fd, etas, edges, return_info=True)
    res["eigs_b2"], res["iters_b2"] = eigs, info["iters"]
    ththmod.sweep_precision("mixed")        # complex64 iteration + complex128 certificate (a 300^2 case: four-row strips)
    try:
        eigs, info = ththmod.eval_sweep(cs_t, tau, fd, etas, edges, return_info=True)
        res["eigs_mx"], res["iters_mx"] = eigs, info["iters"]
        d3, f3, t3, e3 = arc_dynspec(300, 300, seed=9, nimg=6, noise=0.05)
        fd3, tau3 = thth_oracle.fft_axis(t3, 1000.0, 0), thth_oracle.fft_axis(f3, 1.0, 0)
        cs3 = ththmod.conjugate_spectrum(d3 - d3.mean(), 0, pad_value=0.0)
        eigs, info = ththmod.eval_sweep(cs3, tau3, fd3, np.array([0.8, 1.3]) * e3, np.linspace(-fd3.max() / 2, fd3.max() / 2, 300),
                                        return_info=True)
        res["eigs_mx300"], res["iters_mx300"] = eigs, info["iters"]
    finally:
        ththmod.sweep_precision("f64")
    w, V, _ = ththmod.eigvec_sweep(cs_t, tau, fd, etas[1:4], edges)
    res["w_b2"], res["V_b2"] = w, V.cpu().numpy()
    m = ththmod.modeler(CS, tau, fd, etas[2], edges)
    res["recov"], res["model"], res["V1"] = m[2], m[3], m[6]
    res["chisq"] = ththmod.chisq_sweep(dyn, cs_t, tau, fd, etas[1:4], edges, 1.0)
    # round 6: npad = 0 on symmetric axes -- the diagonal back-map (strided sweeps and the per-wavefront segmented scan with its
    # carried rows) with chi^2 taken from its accumulators
    d3, f3, t3, e3 = arc_dynspec(300, 300, seed=9, nimg=6, noise=0.05)
    d3 = d3 - d3.mean()
    fd3, tau3 = thth_oracle.fft_axis(t3, 1000.0, 0), thth_oracle.fft_axis(f3, 1.0, 0)
    chis, cinfo = ththmod.chisq_sweep(d3, ththmod.conjugate_spectrum(d3, 0, pad_value=0.0), tau3, fd3, np.array([0.04, 0.3, 0.8, 1.3, 2.5]) * e3,
                                      np.linspace(-fd3.max() / 2, fd3.max() / 2, 300), 1.0, return_info=True)
    assert cinfo["fused"]
    res["chisq_fused"] = chis
    rng = np.random.default_rng(0)
    a = rng.standard_normal((99, 99)) + 1j * rng.standard_normal((99, 99))
    res["rev_h"] = ththmod.rev_map(a + a.conj().T, tau, fd, etas[2], edges, True)
    res["rev_n"] = ththmod.rev_map(a, tau, fd, etas[2], edges, False)
    np.savez(out_path, **res)
    patch.undo()


if __name__ == "__main__":
    main(sys.argv[1])
