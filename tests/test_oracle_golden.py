"""Pin the CPU oracle against outputs of the unmodified reference
(tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import sspec_oracle as so
from oracle import thth_oracle as to


def phase_align(v, ref):
    return v * np.exp(-1j * np.angle(np.vdot(ref, v)))


# ---------------------------------------------------------------- theta-theta
@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_small_elementwise(golden, tag, k):
    g = golden("thth_small.npz")
    CS, tau, fd = g["CS"], g["tau"], g["fd"]
    eta = g["etas"][k]
    edges = g[f"edges_{tag}"]
    # axes: bit-equal to the reference's fft_axis
    assert np.array_equal(to.fft_axis(g["times"], 1000.0, int(g["npad"])), fd)
    assert np.array_equal(to.fft_axis(g["freqs"], 1.0, int(g["npad"])), tau)
    # gather: bit-equal (a copy times one sqrt)
    assert np.array_equal(to.thth_map(CS, tau, fd, eta, edges), g[f"map_{tag}{k}"])
    assert np.array_equal(to.thth_map(CS, tau, fd, eta, edges, hermetian=False),
                          g[f"mapnh_{tag}{k}"])
    red, edges_red = to.thth_redmap(CS, tau, fd, eta, edges)
    assert np.array_equal(red, g[f"red_{tag}{k}"])
    assert np.array_equal(edges_red, g[f"edgesred_{tag}{k}"])
    if tag == "b":
        assert red.shape[0] < edges.shape[0] - 1      # the crop really happened
    # eigen: same ARPACK call
    assert to.Eval_calc(CS, tau, fd, eta, edges) == pytest.approx(g[f"eval_{tag}{k}"], rel=1e-12)
    # scatter
    np.testing.assert_allclose(to.rev_map(red, tau, fd, eta, edges_red), g[f"rev_{tag}{k}"],
                               rtol=1e-13, atol=0)
    np.testing.assert_allclose(to.rev_map(red, tau, fd, eta, edges_red, hermetian=False),
                               g[f"revnh_{tag}{k}"], rtol=1e-13, atol=0)
    # model (V has an arbitrary global phase; thth2/recov/model are phase free)
    m = to.modeler(CS, tau, fd, eta, edges)
    scale = np.abs(g[f"mod_thth2_{tag}{k}"]).max()
    np.testing.assert_allclose(m[1], g[f"mod_thth2_{tag}{k}"], rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(m[2], g[f"mod_recov_{tag}{k}"], rtol=0,
                               atol=1e-9 * np.abs(g[f"mod_recov_{tag}{k}"]).max())
    np.testing.assert_allclose(m[3], g[f"mod_model_{tag}{k}"], rtol=0,
                               atol=1e-9 * np.abs(g[f"mod_model_{tag}{k}"]).max())
    assert m[5] == pytest.approx(g[f"mod_w_{tag}{k}"], rel=1e-12)
    np.testing.assert_allclose(phase_align(m[6], g[f"mod_V_{tag}{k}"]), g[f"mod_V_{tag}{k}"],
                               rtol=0, atol=1e-8)
    assert to.chisq_calc(g["dyn"], CS, tau, fd, eta, edges, 1.0) == pytest.approx(
        g[f"chisq_{tag}{k}"], rel=1e-9)


def test_min_edges(golden):
    g = golden("thth_small.npz")
    got = to.min_edges(0.4 * g["fd"].max(), g["fd"], g["tau"], float(g["eta_true"]), 2)
    assert np.array_equal(got, g["min_edges"])


def test_medium_checksums(golden):
    from scintools_amd.synth import arc_dynspec
    g = golden("thth_medium.npz")
    dyn, freqs, times, _ = arc_dynspec(int(g["nf"]), int(g["nt"]), seed=int(g["seed"]),
                                       nimg=int(g["nimg"]))
    dyn = dyn - dyn.mean()
    assert np.abs(dyn).sum() == pytest.approx(float(g["dyn_checksum"]), rel=1e-12)
    fd = to.fft_axis(times, 1000.0, 0)
    tau = to.fft_axis(freqs, 1.0, 0)
    CS = to.conjugate_spectrum(dyn, 0)
    for i in (0, 7, 15):
        red, _ = to.thth_redmap(CS, tau, fd, g["etas"][i], g["edges"])
        assert red.shape[0] == g["nred"][i]
        wgt = np.arange(red.size).reshape(red.shape) % 251 + 1
        assert (red * wgt).sum() == pytest.approx(g["csum"][i], rel=1e-9)
        assert np.abs(red).sum() == pytest.approx(g["asum"][i], rel=1e-10)
        assert to.Eval_calc(CS, tau, fd, g["etas"][i], g["edges"]) == pytest.approx(
            g["eigs"][i], rel=1e-10)


def test_tutorial_known_answer(golden):
    """eta ~ 44 s**3 on Sample_Data.npz (thth_intro.rst:101-103) and equality with
    the reference's own single_search on the same chunk."""
    g = golden("thth_sample.npz")
    sel = slice(30, 42)            # the 12 etas around the peak keep this test quick
    etas = g["etas"][sel]
    fd = to.fft_axis(g["time"], 1000.0, int(g["npad"]))
    tau = to.fft_axis(g["freq"], 1.0, int(g["npad"]))
    assert np.array_equal(fd, g["fd"]) and np.array_equal(tau, g["tau"])
    CS = to.conjugate_spectrum(g["chunk"], int(g["npad"]), tau, 0.0)
    eigs = np.array([to.Eval_calc(CS, tau, fd, e, g["edges"]) for e in etas])
    np.testing.assert_allclose(eigs, g["eigs"][sel], rtol=1e-10)
    eigs_i = np.array([to.Eval_calc(np.abs(CS), tau, fd, e, g["edges"]) for e in etas[:3]])
    np.testing.assert_allclose(eigs_i, g["eigs_incoh"][sel][:3], rtol=1e-10)
    # the fit, fed the reference's full eigenvalue curve
    eta_fit, eta_sig, _ = to.fit_eig_peak(g["etas"], g["eigs"], 0.1)
    assert eta_fit == pytest.approx(float(g["eta_fit"]), rel=1e-9)
    assert eta_sig == pytest.approx(float(g["eta_sig"]), rel=1e-6)
    assert abs(eta_fit - 44.0) < 0.1 * 44.0


def test_sim_sweep_with_tau_mask(golden):
    g = golden("sim_sspec.npz")
    dyn = g["dyn"] - np.nanmean(g["dyn"])
    res = to.single_search(dyn, g["freqs"], g["times"], g["sw_etas"], g["sw_edges"],
                           fw=float(g["sw_fw"]), npad=int(g["sw_npad"]), coher=True,
                           tau_mask=float(g["sw_tau_mask"]))
    np.testing.assert_allclose(res[4], g["sw_eigs"], rtol=1e-10)
    assert res[0] == pytest.approx(float(g["sw_eta_fit"]), rel=1e-8)
    assert res[1] == pytest.approx(float(g["sw_eta_sig"]), rel=1e-6)


# ---------------------------------------------------------------- secondary spectrum
SSPEC_CASES = {
    "default": {},
    "prewhite": dict(prewhite=True),
    "full": dict(halve=False),
    "hamming": dict(window="hamming", window_frac=0.25),
    "blackman_pw": dict(window="blackman", window_frac=0.3, prewhite=True),
    "bartlett": dict(window="bartlett", window_frac=0.2),
    "nowindow": dict(window=None),
}


@pytest.mark.parametrize("tag", sorted(SSPEC_CASES))
def test_calc_sspec(golden, tag):
    g = golden("sim_sspec.npz")
    # float64 input: the double-precision answer (same NumPy calls: bit-equal)
    dyn64 = g["dyn"].astype(np.float64)
    fdop, tdel, sec = so.calc_sspec(dyn64, float(g["dt"]), float(g["df"]), **SSPEC_CASES[tag])
    assert np.array_equal(fdop, g[f"fdop_{tag}"])
    assert np.array_equal(tdel, g[f"tdel_{tag}"])
    assert np.array_equal(sec, g[f"sec64_{tag}"])
    # the Simulation's own float32 dyn: NumPy 2 keeps that FFT in single precision
    if f"sec_{tag}" in g.files:
        assert g["dyn"].dtype == np.float32
        sec32 = so.calc_sspec(g["dyn"], float(g["dt"]), float(g["df"]), **SSPEC_CASES[tag])[2]
        assert np.array_equal(sec32, g[f"sec_{tag}"])


def test_calc_sspec_odd_shape(golden):
    g = golden("sim_sspec.npz")
    sub = g["dyn"][:75, :101]
    fdop, tdel, sec = so.calc_sspec(sub, float(g["dt"]), float(g["df"]), prewhite=True)
    assert np.array_equal(sec, g["sub_sec"])
    sec64 = so.calc_sspec(sub.astype(np.float64), float(g["dt"]), float(g["df"]), prewhite=True)[2]
    assert np.array_equal(sec64, g["sub_sec64"])
    assert np.array_equal(fdop, g["sub_fdop"]) and np.array_equal(tdel, g["sub_tdel"])


def test_window_and_acf(golden):
    g = golden("sim_sspec.npz")
    for nt, nf in ((101, 75), (128, 96), (20, 20)):
        cw, sw = so.get_window(nt, nf)
        assert np.array_equal(cw, g[f"win_t_{nt}_{nf}"])
        assert np.array_equal(sw, g[f"win_f_{nt}_{nf}"])
    assert np.array_equal(so.calc_acf(g["dyn"]), g["acf"])


# ---------------------------------------------------------------- phase retrieval
def _align(a, ref):
    """Remove the arbitrary global phase (the eigenvector's) before comparing wavefields."""
    return a * np.exp(-1j * np.angle(np.vdot(ref, a)))


def test_retrieval_chunks_mosaic_gs(golden):
    """single_chunk_retrieval / mosaic / gerchberg_saxton of the oracle against the reference's
    Dynspec.thetatheta_chunks + calc_wavefield + gerchberg_saxton run (first 256 channels of
    Sample_Data.npz).  Wavefields are defined up to one global phase."""
    g = golden("retrieval.npz")
    f = golden("fit_thetatheta.npz")
    n = int(g["nchan"])
    dspec, freq, time = f["dspec"][:n], f["freq"][:n], f["time"]
    cwf, npad = 64, 3
    fref, ththeta, edges = float(g["fref"]), float(g["ththeta"]), g["edges"]
    ncf = n // (cwf // 2) - 1
    chunks = np.zeros((ncf, 1, cwf, time.shape[0]), dtype=complex)
    for cf in range(ncf):
        fs = slice(cf * (cwf // 2), cf * (cwf // 2) + cwf)
        d2 = np.copy(dspec[fs])
        d2 -= np.nanmean(d2)
        fm = freq[fs].mean()
        chunks[cf, 0] = to.single_chunk_retrieval(np.nan_to_num(d2), edges * (fm / fref), time, freq[fs],
                                                  ththeta * (fref / fm) ** 2, npad)
    for cf, key in ((0, "chunk0"), (3, "chunk3")):
        ref = g[key]
        assert np.abs(_align(chunks[cf, 0], ref) - ref).max() <= 1e-7 * np.abs(ref).max()
    wf = to.mosaic(chunks)
    ref = g["wavefield"]
    assert wf.shape == ref.shape
    assert np.abs(_align(wf, ref) - ref).max() <= 1e-7 * np.abs(ref).max()
    gs = to.gerchberg_saxton(wf, dspec, to.fft_axis(freq[: wf.shape[0]], 1.0), niter=2)
    ref = g["wavefield_gs"]
    assert np.abs(_align(gs, ref) - ref).max() <= 1e-7 * np.abs(ref).max()


def test_one_256_channel_chunk_of_the_tutorial_data(golden):
    """The oracle's Eval_calc on ONE 256-channel fitting chunk of the tutorial data (npad = 3: a 1024 x 600 conjugate spectrum,
    1166 default edges) against the reference's own thetatheta_single curve of tests/golden/fit_thetatheta_256.npz
    (make_golden.py fit256: 76 s of the reference for the 52 curvatures; four of them here)."""
    g, f = golden("fit_thetatheta_256.npz"), golden("fit_thetatheta.npz")
    n = int(g["nchan"])
    dspec, freq, time = f["dspec"][:n], f["freq"][:n], f["time"]
    d2 = np.copy(dspec)
    d2 -= np.nanmean(d2)
    npad = int(g["npad"])
    fd, tau = to.fft_axis(time, 1000.0, npad), to.fft_axis(freq, 1.0, npad)
    CS = to.conjugate_spectrum(np.nan_to_num(d2), npad, tau, 0.0)
    edges = g["edges"] * (freq.mean() / float(g["fref"]))
    for i in (0, 17, 34, 51):
        assert to.Eval_calc(CS, tau, fd, g["single_etas"][i], edges) == pytest.approx(g["single_eigs"][i], rel=1e-10)


def test_calc_asymmetry(golden):
    g = golden("retrieval.npz")
    f = golden("fit_thetatheta.npz")
    n = int(g["nchan"])
    dspec, freq, time = f["dspec"][:n], f["freq"][:n], f["time"]
    fref, ththeta, edges = float(g["fref"]), float(g["ththeta"]), g["edges"]
    for cf in range(n // 64):
        fs = slice(cf * 64, (cf + 1) * 64)
        d2 = np.copy(dspec[fs])
        d2 -= np.nanmean(d2)
        fm = freq[fs].mean()
        a = to.calc_asymmetry(np.nan_to_num(d2), edges * (fm / fref), time, freq[fs], ththeta * (fref / fm) ** 2, 3)
        assert a == pytest.approx(g["asymmetry"][cf, 0].real, rel=1e-7, abs=1e-9)


# ---------------------------------------------------------------- arc normalisation (section 8f, rank 3)
from oracle import arcfit_oracle as ao  # noqa: E402


def _same(a, b, **kw):
    np.testing.assert_allclose(np.ma.filled(np.ma.array(a, dtype=float), np.nan), b, equal_nan=True, **kw)


def test_scale_dyn_lambda_and_lamsspec(golden):
    g = golden("arcfit.npz")
    r = ao.calc_sspec_lam(g["dyn"], g["freqs"], float(g["dt"]), float(g["df"]))
    assert np.array_equal(r["lamdyn"], g["lamdyn"])
    assert np.array_equal(r["lam"], g["lam"])
    assert r["dlam"] == float(g["dlam"])
    assert np.array_equal(r["beta"], g["beta"])
    np.testing.assert_allclose(r["lamsspec"], g["lamsspec"], rtol=1e-12, atol=1e-9)


_NORM_CASES = {
    "na": dict(lamsteps=True),
    "nb": dict(lamsteps=True, logsteps=True, numsteps=301, weighted=False, maxnormfac=3, startbin=2, cutmid=4),
    "nc": dict(lamsteps=True, subtract_artefacts=True, powerspec_cut=True, minnormfac=0.3, maxnormfac=2,
               delmax_frac=0.5),
    "nd": dict(lamsteps=False, startbin=3, maxnormfac=3, cutmid=4, eta=130.0),
}


def norm_case(g, tag, fn):
    kw = dict(_NORM_CASES[tag])
    lam = kw["lamsteps"]
    eta = kw.pop("eta", float(g["fa_betaeta"]))
    frac = kw.pop("delmax_frac", None)
    if frac is not None:
        kw["delmax"] = frac * np.max(g["tdel"])
    return fn(g["lamsspec"] if lam else g["sspec"], g["beta"] if lam else g["tdel"], g["tdel"], g["fdop"],
              float(g["freq"]), eta, **kw)


@pytest.mark.parametrize("tag", sorted(_NORM_CASES))
def test_norm_sspec_oracle(golden, tag):
    g = golden("arcfit.npz")
    r = norm_case(g, tag, ao.norm_sspec)
    assert np.array_equal(np.ma.getmaskarray(r["normsspec"]), g[f"{tag}_mask"])
    assert np.array_equal(np.asarray(r["normsspec"].data), g[f"{tag}_norm"], equal_nan=True)
    assert np.array_equal(r["normsspec_fdop"], g[f"{tag}_fdop"])
    assert np.array_equal(r["normsspec_tdel"], g[f"{tag}_tdel"])
    _same(r["powerspectrum"], g[f"{tag}_pow"], rtol=1e-14)
    _same(r["weights"], g[f"{tag}_weights"], rtol=1e-14)
    _same(r["normsspecavg"], g[f"{tag}_avg"], rtol=1e-13)


def test_fit_arc_oracle(golden):
    g = golden("arcfit.npz")
    r = ao.fit_arc(g["lamsspec"], g["beta"], g["tdel"], g["beta"], g["fdop"], float(g["freq"]),
                   lamsteps=True, numsteps=2000)
    s = r["sides"][0]
    assert r["noise"] == pytest.approx(float(g["fa_noise"]), rel=1e-14)
    _same(r["norm"]["normsspecavg"], g["fa_avg"], rtol=1e-13)
    np.testing.assert_allclose(s["eta_array"], g["fa_eta_array"], rtol=1e-14)
    np.testing.assert_allclose(s["spec"], g["fa_spec"], rtol=1e-13)
    np.testing.assert_allclose(s["prob"], g["fa_prob"], rtol=1e-10)
    assert s["eta"] == pytest.approx(float(g["fa_betaeta"]), rel=1e-10)
    assert s["etaerr"] == pytest.approx(float(g["fa_betaetaerr"]), rel=1e-10)
    assert s["etaerr2"] == pytest.approx(float(g["fa_betaetaerr2"]), rel=1e-8)
    r = ao.fit_arc(g["lamsspec"], g["beta"], g["tdel"], g["beta"], g["fdop"], float(g["freq"]),
                   lamsteps=True, numsteps=1500, asymm=True, log_parabola=True, logsteps=True, weighted=True,
                   etamin=40.0, etamax=4000.0, constraint=[100, 2000], nsmooth=7, startbin=4, cutmid=5,
                   delmax=0.8 * np.max(g["tdel"]))
    left, right = r["sides"]
    assert left["eta"] == pytest.approx(float(g["fb_left"]), rel=1e-10)
    assert right["eta"] == pytest.approx(float(g["fb_right"]), rel=1e-10)
    assert left["etaerr"] == pytest.approx(float(g["fb_lefterr"]), rel=1e-10)
    assert right["etaerr"] == pytest.approx(float(g["fb_righterr"]), rel=1e-10)
    np.testing.assert_allclose(left["spec"], g["fb_spec1"], rtol=1e-13)
    np.testing.assert_allclose(right["spec"], g["fb_spec2"], rtol=1e-13)
    np.testing.assert_allclose(right["eta_array"], g["fb_eta_array"], rtol=1e-14)


def test_sim_oracle_is_bit_identical_to_the_reference_simulator(golden):
    """oracle/sim_oracle.py against the reference's own Simulation run: the full 96 x 128 array of
    tests/golden/sim_sspec.npz, and the SHA-256 of the 1024^2 BASELINE-style screen that
    tests/golden/sim_sweep.npz was computed on (make_golden.py::gen_sim_sweep)."""
    from oracle import sim_oracle as so
    g = golden("sim_sspec.npz")
    s = so.Simulation(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.03, freq=1400, dt=30,
                      nx=128, ny=64, nf=96, seed=1234)
    assert s.dyn.dtype == g["dyn"].dtype and np.array_equal(s.dyn, g["dyn"])
    assert np.array_equal(s.freqs, g["freqs"]) and np.array_equal(s.times, g["times"])
    assert s.eta == float(g["sim_eta"]) and s.dt == float(g["dt"]) and s.df == float(g["df"])
    w = golden("sim_sweep.npz")
    big = so.baseline_dynspec(1024, int(w["s1024_seed"]), workers=so.default_workers(8))    # frequencies dealt to worker processes:
    assert so.checksum(big.dyn) == str(w["s1024_sha256"])                                    # the same bits as the loop
    par = so.Simulation(mb2=20, ar=10, psi=0, alpha=5 / 3, inner=0.001, ds=0.01, dlam=0.03, freq=1400, dt=30,
                        nx=128, ny=64, nf=96, seed=1234, workers=3)
    assert np.array_equal(par.dyn, g["dyn"])
    assert big.eta == float(w["s1024_sim_eta"]) and np.array_equal(big.freqs[:2], w["s1024_freqs01"])
