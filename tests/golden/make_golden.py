#!/usr/bin/env python
"""Generate the golden vectors in tests/golden/*.npz by running the UNMODIFIED
reference (/root/reference/scintools) in the build container.

    python tests/golden/make_golden.py

astropy and lmfit are not installable here (no network); they are replaced by
the small stand-ins under tests/golden/refshim (a unit-tracking ndarray
subclass -- see refshim/astropy/units.py for its numerical contract).  Every
number written below is produced by the reference's own functions:
``ththmod.thth_map / thth_redmap / rev_map / modeler / chisq_calc / Eval_calc /
single_search / fft_axis / min_edges``, ``Dynspec.calc_sspec / calc_acf``,
``scint_utils.get_window`` and ``scint_sim.Simulation``.

/root/reference does not exist on the GPU box, so nothing in tests/ reads it
at run time; the tests read the .npz files committed next to this script.
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import matplotlib  # noqa: E402
matplotlib.use("Agg")
import numpy as np  # noqa: E402
import astropy.units as u  # noqa: E402  (the shim)
import scintools.ththmod as thth  # noqa: E402
from scintools.scint_sim import Simulation  # noqa: E402
from scintools.dynspec import Dynspec  # noqa: E402
from scintools.scint_utils import get_window  # noqa: E402
from scintools_amd.synth import arc_dynspec  # noqa: E402

warnings.simplefilter("ignore")


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, keys={sorted(arrs)}")


def V(q):
    """Strip the shim's unit."""
    return np.array(getattr(q, "value", q))


def cs_of(dspec, npad):
    pad = np.pad(dspec, ((0, npad * dspec.shape[0]), (0, npad * dspec.shape[1])),
                 mode="constant", constant_values=dspec.mean())
    return np.fft.fftshift(np.fft.fft2(pad))


# ---------------------------------------------------------------------------
# 1. small analytic arc: every theta-theta function, element by element
# ---------------------------------------------------------------------------
def gen_thth_small():
    dyn, freqs, times, eta_true = arc_dynspec(64, 48, seed=11, nimg=12)
    dyn = dyn - dyn.mean()
    npad = 1
    fd = thth.fft_axis(times * u.s, u.mHz, npad)
    tau = thth.fft_axis(freqs * u.MHz, u.us, npad)
    CS = cs_of(dyn, npad)
    out = dict(dyn=dyn, freqs=freqs, times=times, npad=npad, fd=V(fd), tau=V(tau), CS=CS,
               eta_true=eta_true)
    # (a) edges well inside the CS: no crop; (b) edges reaching past fd.max/2: crop
    edges_a = np.linspace(-0.45 * V(fd).max(), 0.45 * V(fd).max(), 32)
    edges_b = np.linspace(-0.8 * V(fd).max(), 0.8 * V(fd).max(), 40)
    etas = np.array([0.5, 1.0, 1.7]) * eta_true
    out.update(edges_a=edges_a, edges_b=edges_b, etas=etas)
    for tag, edges in (("a", edges_a), ("b", edges_b)):
        for k, eta in enumerate(etas):
            e = eta * u.s**3
            ed = edges * u.mHz
            out[f"map_{tag}{k}"] = thth.thth_map(CS, tau, fd, e, ed)
            out[f"mapnh_{tag}{k}"] = thth.thth_map(CS, tau, fd, e, ed, hermetian=False)
            red, edges_red = thth.thth_redmap(CS, tau, fd, e, ed)
            out[f"red_{tag}{k}"] = red
            out[f"edgesred_{tag}{k}"] = V(edges_red)
            out[f"eval_{tag}{k}"] = thth.Eval_calc(CS, tau, fd, e, ed)
            m = thth.modeler(CS, tau, fd, e, ed)
            out[f"mod_thth2_{tag}{k}"] = m[1]
            out[f"mod_recov_{tag}{k}"] = m[2]
            out[f"mod_model_{tag}{k}"] = m[3]
            out[f"mod_w_{tag}{k}"] = m[5]
            out[f"mod_V_{tag}{k}"] = m[6]
            out[f"rev_{tag}{k}"] = thth.rev_map(red, tau, fd, e, edges_red)
            out[f"revnh_{tag}{k}"] = thth.rev_map(red, tau, fd, e, edges_red, hermetian=False)
            out[f"chisq_{tag}{k}"] = thth.chisq_calc(dyn, CS, tau, fd, e, ed, 1.0)
    out["min_edges"] = V(thth.min_edges(0.4 * fd.max(), fd, tau, eta_true * u.s**3, 2))
    save("thth_small.npz", **out)


# ---------------------------------------------------------------------------
# 2. the tutorial known answer (docs/source/tutorials/thth_intro.rst:250-308)
# ---------------------------------------------------------------------------
def gen_thth_sample():
    d = np.load("/root/reference/scintools/examples/data/ththsims/Sample_Data.npz")
    dspec = np.abs(d["Espec"]) ** 2
    cwf = 64
    chunk = dspec[:cwf] - dspec[:cwf].mean()
    freq = d["f_MHz"][:cwf]
    time = d["t_s"]
    npad = 3
    edges = np.linspace(-0.4, 0.4, 256)
    etas = np.linspace(12.5, 100.0, 100)
    params = [chunk, freq * u.MHz, time * u.s, etas * u.s**3, edges * u.mHz,
              None, False, 0.1, npad, True, 0 * u.us, False]
    eta_fit, eta_sig, fm, tm, eigs = thth.single_search(params)
    params[9] = False
    eta_fit_i, eta_sig_i, _, _, eigs_i = thth.single_search(params)
    fd = thth.fft_axis(time * u.s, u.mHz, npad)
    tau = thth.fft_axis(freq * u.MHz, u.us, npad)
    save("thth_sample.npz", chunk=chunk, freq=freq, time=time, npad=npad, edges=edges,
         etas=etas, eigs=eigs, eta_fit=V(eta_fit), eta_sig=V(eta_sig), fmean=V(fm), tmean=V(tm),
         eigs_incoh=eigs_i, eta_fit_incoh=V(eta_fit_i), eta_sig_incoh=V(eta_sig_i),
         fd=V(fd), tau=V(tau))


# ---------------------------------------------------------------------------
# 3. reference Simulation -> Dynspec.calc_sspec / calc_acf, and a sweep with tau_mask
# ---------------------------------------------------------------------------
def gen_sim():
    sim = Simulation(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.03,
                     freq=1400, dt=30, nx=128, ny=64, nf=96, seed=1234, lamsteps=False)
    dyn = Dynspec(dyn=sim, process=False, verbose=False)
    out = dict(dyn=np.array(dyn.dyn), dt=dyn.dt, df=dyn.df, freqs=dyn.freqs, times=dyn.times,
               sim_eta=sim.eta)
    for tag, kw in (("default", {}),
                    ("prewhite", dict(prewhite=True)),
                    ("full", dict(halve=False)),
                    ("hamming", dict(window="hamming", window_frac=0.25)),
                    ("blackman_pw", dict(window="blackman", window_frac=0.3, prewhite=True)),
                    ("bartlett", dict(window="bartlett", window_frac=0.2)),
                    ("nowindow", dict(window=None))):
        fdop, tdel, sec = dyn.calc_sspec(return_sspec=True, **kw)
        if tag in ("default", "prewhite"):
            out[f"sec_{tag}"] = sec        # float32-input path (see below)
        out[f"fdop_{tag}"] = fdop
        out[f"tdel_{tag}"] = tdel
        # Simulation's dyn is float32 and NumPy 2 keeps the FFT in single precision for it;
        # the same call on a float64 copy pins the double-precision answer
        out[f"sec64_{tag}"] = dyn.calc_sspec(input_dyn=np.array(dyn.dyn, dtype=np.float64), **kw)[2]
    # odd, non-power-of-two shape through input_dyn
    sub = np.array(dyn.dyn[:75, :101])
    fdop, tdel, sec = dyn.calc_sspec(input_dyn=sub, prewhite=True)
    out.update(sub_sec=sec, sub_fdop=fdop, sub_tdel=tdel)
    out["sub_sec64"] = dyn.calc_sspec(input_dyn=sub.astype(np.float64), prewhite=True)[2]
    dyn.calc_acf()
    out["acf"] = dyn.acf
    for n in ((101, 75), (128, 96), (20, 20)):
        cw, sw = get_window(n[0], n[1], window="hanning", frac=0.1)
        out[f"win_t_{n[0]}_{n[1]}"] = cw
        out[f"win_f_{n[0]}_{n[1]}"] = sw
    # theta-theta sweep on the simulated chunk, with a delay mask and npad=1
    d2 = np.array(dyn.dyn) - np.nanmean(dyn.dyn)
    npad = 1
    fd = thth.fft_axis(dyn.times * u.s, u.mHz, npad)
    edges = np.linspace(-V(fd).max() / 2, V(fd).max() / 2, 64)
    etas = np.geomspace(0.25, 4.0, 24) * sim.eta
    params = [d2, dyn.freqs * u.MHz, dyn.times * u.s, etas * u.s**3, edges * u.mHz,
              None, False, 0.3, npad, True, 0.05 * u.us, False]
    eta_fit, eta_sig, fm, tm, eigs = thth.single_search(params)
    out.update(sw_edges=edges, sw_etas=etas, sw_eigs=eigs, sw_eta_fit=V(eta_fit),
               sw_eta_sig=V(eta_sig), sw_npad=npad, sw_tau_mask=0.05, sw_fw=0.3)
    save("sim_sspec.npz", **out)


# ---------------------------------------------------------------------------
# 4. a medium case: N = 255 eigenvalues + matrix checksums (keeps the file small)
# ---------------------------------------------------------------------------
# NOTE: `modeler` calls eigsh WITHOUT v0, and ARPACK then draws its start vector from the process's RNG state: the
# modeler fixtures (thth_medium.npz, retrieval.npz) reproduce bit for bit only when generated in a FRESH process
# (`python make_golden.py medium`, `python make_golden.py retrieval`, one set per process); generated after other sets in one process
# they differ at 1e-15 / by the eigenvector's global phase.  The tests compare those arrays at 1e-9 / modulo the phase.
def gen_thth_medium():
    dyn, freqs, times, eta_true = arc_dynspec(256, 256, seed=5, nimg=48)
    dyn = dyn - dyn.mean()
    npad = 0
    fd = thth.fft_axis(times * u.s, u.mHz, npad)
    tau = thth.fft_axis(freqs * u.MHz, u.us, npad)
    CS = cs_of(dyn, npad)
    edges = np.linspace(-V(fd).max() / 2, V(fd).max() / 2, 256)
    etas = np.geomspace(0.5, 2.0, 16) * eta_true
    eigs = np.zeros(len(etas))
    nred = np.zeros(len(etas), int)
    csum = np.zeros(len(etas), complex)
    asum = np.zeros(len(etas))
    for i, eta in enumerate(etas):
        red, _ = thth.thth_redmap(CS, tau, fd, eta * u.s**3, edges * u.mHz)
        nred[i] = red.shape[0]
        # position-weighted checksum: catches a transposed or shifted gather
        wgt = np.arange(red.size).reshape(red.shape) % 251 + 1
        csum[i] = (red * wgt).sum()
        asum[i] = np.abs(red).sum()
        eigs[i] = thth.Eval_calc(CS, tau, fd, eta * u.s**3, edges * u.mHz)
    m = thth.modeler(CS, tau, fd, eta_true * u.s**3, edges * u.mHz)
    save("thth_medium.npz", seed=5, nf=256, nt=256, nimg=48, npad=npad, edges=edges, etas=etas,
         eigs=eigs, nred=nred, csum=csum, asum=asum, eta_true=eta_true,
         mod_w=m[5], mod_V=m[6], mod_model_sum=np.abs(m[3]).sum(),
         mod_model_row=m[3][17], mod_recov_row=m[2][100],
         dyn_checksum=np.abs(dyn).sum())


# ---------------------------------------------------------------------------
# 5. the Dynspec path of the tutorial (docs/source/tutorials/dynspec_thth.rst:86-170)
# ---------------------------------------------------------------------------
def gen_fit_thetatheta():
    from scintools.dynspec import BasicDyn
    d = np.load("/root/reference/scintools/examples/data/ththsims/Sample_Data.npz")
    dspec = np.abs(d["Espec"]) ** 2
    freq, tme = d["f_MHz"], d["t_s"]
    b = BasicDyn(name="Sample Data", header=["Sample Data"], times=tme, freqs=freq, dyn=dspec,
                 nsub=tme.shape[0], nchan=freq.shape[0], dt=(tme[1] - tme[0]), df=(freq[1] - freq[0]))
    dyn = Dynspec(dyn=b, process=False, verbose=False)
    dyn.prep_thetatheta(verbose=False, cwf=64, edges_lim=.3, eta_min=30 * u.s**3, eta_max=50 * u.s**3)
    etas0, eigs0, popt0 = dyn.thetatheta_single(cf=0, ct=0, plot=False, arrays=True)
    dyn.fit_thetatheta(verbose=False)
    save("fit_thetatheta.npz", dspec=dspec, freq=freq, time=tme, dt=b.dt, df=b.df,
         edges=V(dyn.edges), neta=dyn.neta, npad=dyn.npad, fw=dyn.fw, fref=V(dyn.fref),
         eta_min=V(dyn.eta_min), eta_max=V(dyn.eta_max), cwf=dyn.cwf, cwt=dyn.cwt,
         single_etas=V(etas0), single_eigs=eigs0, single_popt=np.array(popt0),
         eta_evo=V(dyn.eta_evo), eta_evo_err=V(dyn.eta_evo_err), f0s=V(dyn.f0s), t0s=V(dyn.t0s),
         ththeta=V(dyn.ththeta), ththetaerr=V(dyn.ththetaerr))


def gen_fit_thetatheta_256():
    """The same tutorial data with ONE 256-channel fitting chunk (the first 256 of its 448 channels; npad = 3: a 1024 x 600
    conjugate spectrum, the default edges of min_edges): parity of thetatheta_single / fit_thetatheta beyond the 64-channel
    chunks of the tutorial recipe.  Also records how long the reference took."""
    import time
    from scintools.dynspec import BasicDyn
    d = np.load("/root/reference/scintools/examples/data/ththsims/Sample_Data.npz")
    nchan = 256
    dspec = (np.abs(d["Espec"]) ** 2)[:nchan]
    freq, tme = d["f_MHz"][:nchan], d["t_s"]
    b = BasicDyn(name="Sample Data", header=["Sample Data"], times=tme, freqs=freq, dyn=dspec,
                 nsub=tme.shape[0], nchan=freq.shape[0], dt=(tme[1] - tme[0]), df=(freq[1] - freq[0]))
    dyn = Dynspec(dyn=b, process=False, verbose=False)
    dyn.prep_thetatheta(verbose=False, cwf=256, edges_lim=.3, eta_min=30 * u.s**3, eta_max=50 * u.s**3)
    t0 = time.perf_counter()
    etas0, eigs0, popt0 = dyn.thetatheta_single(cf=0, ct=0, plot=False, arrays=True)
    t1 = time.perf_counter()
    dyn.fit_thetatheta(verbose=False)
    t2 = time.perf_counter()
    save("fit_thetatheta_256.npz", nchan=nchan, edges=V(dyn.edges), neta=dyn.neta, npad=dyn.npad, fw=dyn.fw, fref=V(dyn.fref),
         eta_min=V(dyn.eta_min), eta_max=V(dyn.eta_max), cwf=dyn.cwf, cwt=dyn.cwt,
         single_etas=V(etas0), single_eigs=eigs0, single_popt=np.array(popt0),
         eta_evo=V(dyn.eta_evo), eta_evo_err=V(dyn.eta_evo_err), f0s=V(dyn.f0s), t0s=V(dyn.t0s),
         ththeta=V(dyn.ththeta), ththetaerr=V(dyn.ththetaerr),
         reference_seconds=np.array([t1 - t0, t2 - t1]), host_cores=os.cpu_count())


# ---------------------------------------------------------------------------
# 6. phase retrieval: thetatheta_chunks -> mosaic -> Gerchberg-Saxton (dynspec.py:1765-1875)
# ---------------------------------------------------------------------------
def gen_retrieval():
    from scintools.dynspec import BasicDyn
    d = np.load("/root/reference/scintools/examples/data/ththsims/Sample_Data.npz")
    nchan = 256
    dspec = (np.abs(d["Espec"]) ** 2)[:nchan]
    freq, tme = d["f_MHz"][:nchan], d["t_s"]
    b = BasicDyn(name="Sample Data", header=["Sample Data"], times=tme, freqs=freq, dyn=dspec,
                 nsub=tme.shape[0], nchan=freq.shape[0], dt=(tme[1] - tme[0]), df=(freq[1] - freq[0]))
    dyn = Dynspec(dyn=b, process=False, verbose=False)
    dyn.prep_thetatheta(verbose=False, cwf=64, edges_lim=.3, eta_min=30 * u.s**3, eta_max=50 * u.s**3,
                        nedge=128)
    dyn.fit_thetatheta(verbose=False)
    dyn.calc_wavefield()
    wavefield = np.array(dyn.wavefield)
    chunks = np.array(dyn.chunks)
    dyn.gerchberg_saxton(niter=2)
    dyn.calc_asymmetry()
    save("retrieval.npz", asymmetry=np.array(dyn.asymmetry), nchan=nchan, edges=V(dyn.edges), neta=dyn.neta, fref=V(dyn.fref),
         ththeta=V(dyn.ththeta), ththetaerr=V(dyn.ththetaerr), eta_evo=V(dyn.eta_evo),
         chunk0=chunks[0, 0], chunk3=chunks[3, 0],
         wavefield=wavefield, wavefield_gs=np.array(dyn.wavefield))


# ---------------------------------------------------------------------------
# 7. psrflux text I/O (dynspec.py:144-258, 330-376): a small synthetic file, as the reference parses it
# ---------------------------------------------------------------------------
def gen_psrflux():
    rng = np.random.default_rng(12)
    nsub, nchan = 24, 40
    path = os.path.join(HERE, "synthetic.dynspec")
    t_min = 0.05 + 0.5 * np.arange(nsub)        # minutes, leading edges
    t_min[0] = t_min[1] - 0.12                  # a short first sub-integration (gets removed)
    freqs = 1500.0 - 0.78125 * np.arange(nchan)  # descending, as psrflux writes them
    flux = rng.standard_normal((nsub, nchan)) + 5
    with open(path, "w") as fh:
        fh.write("# Dynamic spectrum computed by psrflux\n# Data file: synthetic\n# MJD0: 58000.25\n")
        fh.write("# Data columns:\n# isub ichan time(min) freq(MHz) flux flux_err\n")
        for i in range(nsub):
            for j in range(nchan):
                fh.write(f"{i:4d} {j:4d} {t_min[i]:10.4f} {freqs[j]:12.6f} {flux[i, j]:+e} {0.1:+e}\n")
    d = Dynspec(filename=path, verbose=False, process=False)
    out = dict(dyn=np.array(d.dyn), times=d.times, freqs=d.freqs, nchan=d.nchan, nsub=d.nsub, bw=d.bw,
               df=d.df, freq=d.freq, dt=d.dt, tobs=d.tobs, mjd=d.mjd, nheader=len(d.header))
    wpath = os.path.join(HERE, "_rewritten.dynspec")
    d.write_file(filename=wpath, verbose=False, note="roundtrip")
    d2 = Dynspec(filename=wpath, verbose=False, process=False)
    os.remove(wpath)
    out.update(rt_dyn=np.array(d2.dyn), rt_times=d2.times, rt_freqs=d2.freqs, rt_mjd=d2.mjd)
    save("psrflux.npz", **out)


# ---------------------------------------------------------------------------
# 8. arc normalisation: scale_dyn('lambda') -> calc_sspec(lamsteps) -> fit_arc / norm_sspec
#    (dynspec.py:3928-3959, 970-1346, 1920-2183)
# ---------------------------------------------------------------------------
def gen_arcfit():
    sim = Simulation(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.25,
                     freq=1400, dt=30, nx=256, ny=64, nf=192, seed=77, lamsteps=False)
    dyn = Dynspec(dyn=sim, process=False, verbose=False)
    dyn.dyn = np.array(dyn.dyn, dtype=np.float64)       # float64 input (see gen_sim)
    out = dict(dyn=dyn.dyn, dt=dyn.dt, df=dyn.df, freqs=dyn.freqs, times=dyn.times, freq=dyn.freq,
               sim_eta=sim.eta)
    dyn.calc_sspec(lamsteps=True)
    out.update(lamdyn=dyn.lamdyn, lam=dyn.lam, dlam=dyn.dlam, beta=dyn.beta, lamsspec=dyn.lamsspec,
               fdop=dyn.fdop, tdel=dyn.tdel)

    def grab(tag, d, keys2d=True):
        out[f"{tag}_avg"] = np.ma.filled(np.ma.array(d.normsspecavg, dtype=float), np.nan)
        out[f"{tag}_fdop"] = np.array(d.normsspec_fdop)
        out[f"{tag}_tdel"] = np.array(d.normsspec_tdel)
        out[f"{tag}_pow"] = np.ma.filled(np.ma.array(d.powerspectrum, dtype=float), np.nan)
        out[f"{tag}_weights"] = np.array(d.weights)
        if keys2d:
            out[f"{tag}_norm"] = np.array(d.normsspec.data)
            out[f"{tag}_mask"] = np.array(np.ma.getmaskarray(d.normsspec))

    # (a) the default Hough-style curvature search in wavelength steps
    dyn.fit_arc(lamsteps=True, numsteps=2000)
    grab("fa", dyn, keys2d=False)
    out.update(fa_betaeta=dyn.betaeta, fa_betaetaerr=dyn.betaetaerr, fa_betaetaerr2=dyn.betaetaerr2,
               fa_noise=dyn.noise, fa_eta_array=dyn.eta_array, fa_spec=dyn.norm_sspec_avg,
               fa_prob=dyn.prob_eta_peak)
    # (b) asymmetric, log-parabola, log-spaced, weighted, with bounds and a constraint
    dyn.fit_arc(lamsteps=True, numsteps=1500, asymm=True, log_parabola=True, logsteps=True,
                weighted=True, etamin=40.0, etamax=4000.0, constraint=[100, 2000], nsmooth=7,
                startbin=4, cutmid=5, delmax=0.8 * np.max(dyn.tdel))
    out.update(fb_left=dyn.betaeta_left, fb_right=dyn.betaeta_right,
               fb_lefterr=dyn.betaetaerr_left, fb_righterr=dyn.betaetaerr_right,
               fb_spec1=dyn.norm_sspec_avg1, fb_spec2=dyn.norm_sspec_avg2, fb_eta_array=dyn.eta_array)
    # (c) norm_sspec variants, 2-D output included
    dyn.norm_sspec(eta=out["fa_betaeta"], lamsteps=True, plot=False)
    grab("na", dyn)
    dyn.norm_sspec(eta=out["fa_betaeta"], lamsteps=True, plot=False, logsteps=True, numsteps=301,
                   weighted=False, maxnormfac=3, startbin=2, cutmid=4)
    grab("nb", dyn)
    dyn.norm_sspec(eta=out["fa_betaeta"], lamsteps=True, plot=False, subtract_artefacts=True,
                   powerspec_cut=True, minnormfac=0.3, maxnormfac=2, delmax=0.5 * np.max(dyn.tdel))
    grab("nc", dyn)
    # (d) frequency-step spectrum (eta given in the units norm_sspec converts from, dynspec.py:2033-2038)
    dyn.calc_sspec()
    out["sspec"] = dyn.sspec
    dyn.norm_sspec(eta=130.0, lamsteps=False, plot=False, startbin=3, maxnormfac=3, cutmid=4)
    grab("nd", dyn)
    # (e) prep_thetatheta without curvature bounds: they come from fit_arc (dynspec.py:1458-1473)
    from scintools.dynspec import BasicDyn
    d = np.load("/root/reference/scintools/examples/data/ththsims/Sample_Data.npz")
    dspec = np.abs(d["Espec"]) ** 2
    freq, tme = d["f_MHz"], d["t_s"]
    b = BasicDyn(name="Sample Data", header=["Sample Data"], times=tme, freqs=freq, dyn=dspec,
                 nsub=tme.shape[0], nchan=freq.shape[0], dt=(tme[1] - tme[0]), df=(freq[1] - freq[0]))
    sd = Dynspec(dyn=b, process=False, verbose=False)
    sd.prep_thetatheta(verbose=False, cwf=64, edges_lim=.3)
    out.update(pt_eta_min=V(sd.eta_min), pt_eta_max=V(sd.eta_max), pt_neta=sd.neta, pt_betaeta=sd.betaeta,
               pt_betaetaerr=sd.betaetaerr, pt_betaetaerr2=sd.betaetaerr2, pt_edges=V(sd.edges))
    save("arcfit.npz", **out)


# ---------------------------------------------------------------------------
# 9. BASELINE-style Simulation screens at 1024^2 and 2048^2: the reference's Eval_calc over the
#    whole geomspace(0.25, 4) * sim.eta range (flat parts of the curve included: that is where
#    lambda_2 / lambda_1 -> 1 stresses a Lanczos stopping rule).  The dynamic spectra are not
#    stored (4-16 MB each): oracle/sim_oracle.py regenerates them bit for bit, pinned by SHA-256.
# ---------------------------------------------------------------------------
def gen_sim_sweep():
    import hashlib
    out = {}
    for size, seed, neta in ((1024, 1, 96), (2048, 2, 32)):
        sim = Simulation(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.25, freq=1400, dt=30,
                         nx=size, ny=128, nf=size, seed=seed, lamsteps=False)
        dyn = np.array(sim.dyn, dtype=np.float64)
        d2 = dyn - dyn.mean()
        fd = thth.fft_axis(sim.times * u.s, u.mHz, 0)
        tau = thth.fft_axis(sim.freqs * u.MHz, u.us, 0)
        CS = cs_of(d2, 0)
        edges = np.linspace(-V(fd).max() / 2, V(fd).max() / 2, size)
        etas = np.geomspace(0.25, 4.0, neta) * sim.eta
        eigs = np.array([thth.Eval_calc(CS, tau, fd, e * u.s**3, edges * u.mHz) for e in etas])
        tag = f"s{size}"
        out.update({f"{tag}_seed": seed, f"{tag}_etas": etas, f"{tag}_eigs": eigs, f"{tag}_sim_eta": sim.eta,
                    f"{tag}_sha256": hashlib.sha256(np.ascontiguousarray(sim.dyn).tobytes()).hexdigest(),
                    f"{tag}_freqs01": np.array(sim.freqs[:2]), f"{tag}_dt": sim.dt})
        print(tag, "eta", sim.eta, "peak at", etas[np.argmax(eigs)] / sim.eta)
    save("sim_sweep.npz", **out)


# ---------------------------------------------------------------------------
# 10. The headline size: a reference Simulation screen at 4096^2 (seed 3), seventeen curvatures across
#     geomspace(0.25, 4, 256) * sim.eta (every 16th and the last: both flat ends, lambda_2 / lambda_1 -> 0.99, the peak and
#     the slopes) through the reference's own Eval_calc.  Round 5 pinned five of them (indices 0 / 64 / 128 / 192 / 255, a
#     subset of these).  ~25 min on 8 cores (3 min of it the simulator).
# ---------------------------------------------------------------------------
def gen_sim_sweep_4096():
    import hashlib
    size, seed = 4096, 3
    sim = Simulation(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.25, freq=1400, dt=30,
                     nx=size, ny=128, nf=size, seed=seed, lamsteps=False)
    dyn = np.array(sim.dyn, dtype=np.float64)
    d2 = dyn - dyn.mean()
    fd = thth.fft_axis(sim.times * u.s, u.mHz, 0)
    tau = thth.fft_axis(sim.freqs * u.MHz, u.us, 0)
    CS = cs_of(d2, 0)
    edges = np.linspace(-V(fd).max() / 2, V(fd).max() / 2, size)
    idx = np.append(np.arange(0, 256, 16), 255)
    etas = (np.geomspace(0.25, 4.0, 256) * sim.eta)[idx]
    eigs = np.array([thth.Eval_calc(CS, tau, fd, e * u.s**3, edges * u.mHz) for e in etas])
    save("sim_sweep_4096.npz", seed=seed, idx=idx, etas=etas, eigs=eigs, sim_eta=sim.eta,
         sha256=hashlib.sha256(np.ascontiguousarray(sim.dyn).tobytes()).hexdigest(),
         freqs01=np.array(sim.freqs[:2]), dt=sim.dt)
    print("s4096 eta", sim.eta, "eigs", eigs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "sample", "sim", "medium", "fit", "retrieval", "psrflux", "arcfit"]
    if "small" in which:
        gen_thth_small()
    if "sample" in which:
        gen_thth_sample()
    if "sim" in which:
        gen_sim()
    if "medium" in which:
        gen_thth_medium()
    if "fit" in which:
        gen_fit_thetatheta()
    if "retrieval" in which:
        gen_retrieval()
    if "psrflux" in which:
        gen_psrflux()
    if "arcfit" in which:
        gen_arcfit()
    if "simsweep" in which:        # ~5 min: not in the default list
        gen_sim_sweep()
    if "simsweep4096" in which:    # ~15 min: not in the default list
        gen_sim_sweep_4096()
    if "fit256" in which:          # one 256-channel chunk of the tutorial data: minutes, not in the default list
        gen_fit_thetatheta_256()
