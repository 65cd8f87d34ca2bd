"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors made from the reference.  Run with `pytest -m gpu` on an MI355X.

Tolerances:
  gather            bit-equal values, hence identical index maps
  FFT / CS          |err| <= 1e-12 * max|X| (float64 FFT, different radix order)
  sspec             1e-8 dB absolute where the power is above the rounding floor
  eigenvalue |w|    rtol 1e-9 against ARPACK eigsh
  V                 1 - |<V_gpu, V_ref>| <= 1e-9
  rev_map / model   rtol 1e-9 of the array maximum (float64 LDS-atomic summation order)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def thth():
    from scintools_amd import ththmod
    from scintools_amd.device import require_gpu
    require_gpu()
    return ththmod


@pytest.fixture(scope="module")
def to():
    from oracle import thth_oracle
    return thth_oracle


def _native_loaded():
    with open("/proc/self/maps") as fh:
        return "libscint_hip.so" in fh.read()


def test_native_library_is_loaded(thth):
    assert _native_loaded()


# ------------------------------------------------------------------ FFT
@pytest.mark.parametrize("shape", [(2, 16), (4, 32), (32, 64), (64, 128), (128, 256), (256, 512),
                                   (512, 1024), (1024, 2048), (16, 4096), (8, 8192), (4, 16384),
                                   (2048, 16), (4096, 64), (8192, 32),
                                   # not powers of two: chirp-z on one or both axes
                                   (2, 1), (3, 5), (7, 16), (16, 7), (75, 101), (128, 96), (96, 128),
                                   (256, 600), (600, 256), (150, 1024), (1000, 30), (30, 5000),
                                   (4500, 12)])
def test_fft2_matches_numpy(thth, shape):
    import ctypes
    import torch
    from scintools_amd import _lib
    from scintools_amd.device import empty, ptr, stream_ptr, to_device
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    lib = _lib.load()
    need = ctypes.c_size_t()
    _lib.check(lib.scint_fft2_workspace_bytes(shape[0], shape[1], ctypes.byref(need)))
    ws = empty((need.value,), torch.uint8)
    xt = to_device(x, torch.complex128)
    out = empty(shape, torch.complex128)
    _lib.check(lib.scint_fft2(ptr(xt), ptr(out), shape[0], shape[1], ptr(ws), ws.numel(), stream_ptr()))
    got = out.cpu().numpy()
    ref = np.fft.fft2(x)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("nf,nt,npad", [(64, 64, 0), (64, 32, 1), (32, 128, 3), (256, 256, 0), (16, 16, 0),
                                        (64, 48, 1), (64, 150, 3), (75, 101, 0), (50, 20, 2)])
@pytest.mark.parametrize("coher", [True, False])
def test_conjugate_spectrum(thth, to, nf, nt, npad, coher):
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, _ = arc_dynspec(nf, nt, seed=nf + nt + npad, nimg=8)
    tau = to.fft_axis(freqs, 1.0, npad)
    mask = 2.5 * (tau[1] - tau[0])
    ref = to.conjugate_spectrum(dyn, npad, tau, mask)
    if not coher:
        ref = np.abs(ref)
    got = thth.conjugate_spectrum(dyn, npad, tau, mask, coher).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.all(got[np.abs(tau) < mask] == 0)


# ------------------------------------------------------------------ gather
@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_gather_bit_exact_vs_reference_golden(thth, golden, tag, k):
    g = golden("thth_small.npz")
    CS, tau, fd, eta, edges = g["CS"], g["tau"], g["fd"], g["etas"][k], g[f"edges_{tag}"]
    assert np.array_equal(thth.thth_map(CS, tau, fd, eta, edges), g[f"map_{tag}{k}"])
    assert np.array_equal(thth.thth_map(CS, tau, fd, eta, edges, hermetian=False), g[f"mapnh_{tag}{k}"])
    red, edges_red = thth.thth_redmap(CS, tau, fd, eta, edges)
    assert np.array_equal(red, g[f"red_{tag}{k}"])
    assert np.array_equal(np.asarray(edges_red), g[f"edgesred_{tag}{k}"])


@pytest.mark.parametrize("n,nedge,seed", [(256, 256, 5), (512, 300, 9), (1024, 1024, 2)])
def test_gather_bit_exact_vs_oracle(thth, to, n, nedge, seed):
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(n, n, seed=seed, nimg=32)
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    CS = to.conjugate_spectrum(dyn - dyn.mean(), 0)
    edges = np.linspace(-0.7 * fd.max(), 0.7 * fd.max(), nedge)
    cs_dev = thth.to_device(CS)
    for eta in eta_true * np.array([0.4, 1.0, 2.3]):
        ref, eref = to.thth_redmap(CS, tau, fd, eta, edges)
        got, egot = thth.thth_redmap(cs_dev, tau, fd, eta, edges)
        assert got.shape == ref.shape
        mism = np.count_nonzero(got != ref)
        assert mism == 0, f"{mism} mismatched pixels of {ref.size}"
        assert np.array_equal(np.asarray(egot), eref)
    refnh = to.thth_map(CS, tau, fd, eta_true, edges, hermetian=False)
    assert np.array_equal(thth.thth_map(cs_dev, tau, fd, eta_true, edges, hermetian=False), refnh)


# ------------------------------------------------------------------ eigen
def test_eval_sweep_vs_reference_golden_medium(thth, to, golden):
    from scintools_amd.synth import arc_dynspec
    g = golden("thth_medium.npz")
    dyn, freqs, times, _ = arc_dynspec(int(g["nf"]), int(g["nt"]), seed=int(g["seed"]), nimg=int(g["nimg"]))
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    eigs, info = thth.eval_sweep(cs, tau, fd, g["etas"], g["edges"], return_info=True)
    assert np.array_equal(info["N"], g["nred"])
    assert np.all(info["status"] == 0)
    np.testing.assert_allclose(eigs, g["eigs"], rtol=1e-9)
    # batching must not change a single bit
    eigs1 = thth.eval_sweep(cs, tau, fd, g["etas"], g["edges"], batch=1)
    eigs5 = thth.eval_sweep(cs, tau, fd, g["etas"], g["edges"], batch=5)
    assert np.array_equal(eigs, eigs1) and np.array_equal(eigs, eigs5)


def test_eval_sweep_tutorial_known_answer(thth, to, golden):
    """Sample_Data chunk (thth_intro.rst:250-308): curve equals the reference's, peak ~44 s^3."""
    g = golden("thth_sample.npz")
    CS = to.conjugate_spectrum(g["chunk"], int(g["npad"]), g["tau"], 0.0)   # 256 x 600: host FFT
    eigs = thth.eval_sweep(CS, g["tau"], g["fd"], g["etas"], g["edges"])
    np.testing.assert_allclose(eigs, g["eigs"], rtol=1e-9)
    eigs_i = thth.eval_sweep(np.abs(CS), g["tau"], g["fd"], g["etas"], g["edges"])
    np.testing.assert_allclose(eigs_i, g["eigs_incoh"], rtol=1e-9)
    eta_fit, eta_sig, _ = thth.fit_eig_peak(g["etas"], eigs, 0.1)
    assert eta_fit == pytest.approx(float(g["eta_fit"]), rel=1e-6)
    assert abs(eta_fit - 44.0) < 4.4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_eval_calc_small_golden(thth, golden, tag):
    g = golden("thth_small.npz")
    for k in range(3):
        got = thth.Eval_calc(g["CS"], g["tau"], g["fd"], g["etas"][k], g[f"edges_{tag}"])
        assert got == pytest.approx(float(g[f"eval_{tag}{k}"]), rel=1e-9)


def test_eigh_top_against_lapack(thth):
    import torch
    from scintools_amd.ththmod import _eigh_top_dev
    rng = np.random.default_rng(3)
    for n in (2, 3, 17, 64, 257, 700):
        a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        a = a + a.conj().T
        np.fill_diagonal(a, 0)
        w, V, iters = _eigh_top_dev(thth.to_device(a))
        wr, Vr = np.linalg.eigh(a)
        assert w == pytest.approx(wr[-1], rel=1e-10)
        V = V.cpu().numpy()
        assert abs(np.linalg.norm(V) - 1) < 1e-12
        assert 1 - abs(np.vdot(Vr[:, -1], V)) <= 1e-9
        assert np.linalg.norm(a @ V - w * V) <= 1e-8 * abs(w)


# ------------------------------------------------------------------ scatter / model
@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_rev_map_vs_reference_golden(thth, golden, tag, k):
    g = golden("thth_small.npz")
    tau, fd, eta = g["tau"], g["fd"], g["etas"][k]
    red, edges_red = g[f"red_{tag}{k}"], g[f"edgesred_{tag}{k}"]
    for herm, key in ((True, "rev"), (False, "revnh")):
        ref = g[f"{key}_{tag}{k}"]
        got = thth.rev_map(red, tau, fd, eta, edges_red, hermetian=herm)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.array_equal(got == 0, ref == 0)      # same empty bins


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_modeler_and_chisq_vs_reference_golden(thth, golden, tag, k):
    g = golden("thth_small.npz")
    CS, tau, fd, eta, edges = g["CS"], g["tau"], g["fd"], g["etas"][k], g[f"edges_{tag}"]
    red, thth2, recov, model, edges_red, w, V = thth.modeler(CS, tau, fd, eta, edges)
    assert np.array_equal(red, g[f"red_{tag}{k}"])
    assert np.array_equal(np.asarray(edges_red), g[f"edgesred_{tag}{k}"])
    assert w == pytest.approx(float(g[f"mod_w_{tag}{k}"]), rel=1e-9)
    Vr = g[f"mod_V_{tag}{k}"]
    assert 1 - abs(np.vdot(Vr, V)) <= 1e-9
    for got, key in ((thth2, "mod_thth2"), (recov, "mod_recov"), (model, "mod_model")):
        ref = g[f"{key}_{tag}{k}"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max(), key
    chi = thth.chisq_calc(g["dyn"], CS, tau, fd, eta, edges, 1.0)
    assert chi == pytest.approx(float(g[f"chisq_{tag}{k}"]), rel=1e-9)
    mask = np.zeros(g["dyn"].shape, bool)
    mask[::2, 1::3] = True
    ref_model = g[f"mod_model_{tag}{k}"][: mask.shape[0], : mask.shape[1]]
    chi_m = thth.chisq_calc(g["dyn"], CS, tau, fd, eta, edges, 2.5, mask=mask)
    assert chi_m == pytest.approx(np.sum((ref_model - g["dyn"])[mask] ** 2) / 2.5, rel=1e-9)


def test_single_search_end_to_end_vs_reference(thth, golden):
    """Whole chunk search on the GPU (256 x 600 chirp-z FFT, sweep, fit) against the
    reference's own single_search on the tutorial chunk and on the simulated chunk."""
    g = golden("thth_sample.npz")
    params = [g["chunk"], g["freq"], g["time"], g["etas"], g["edges"], None, False, 0.1,
              int(g["npad"]), True, 0.0, False]
    eta_fit, eta_sig, fm, tm, eigs = thth.single_search(params)
    np.testing.assert_allclose(eigs, g["eigs"], rtol=1e-9)
    assert float(eta_fit) == pytest.approx(float(g["eta_fit"]), rel=1e-6)
    assert float(eta_sig) == pytest.approx(float(g["eta_sig"]), rel=1e-4)
    assert float(fm) == pytest.approx(float(g["fmean"])) and float(tm) == pytest.approx(float(g["tmean"]))
    params[9] = False
    res = thth.single_search(params)
    np.testing.assert_allclose(res[4], g["eigs_incoh"], rtol=1e-9)
    assert float(res[0]) == pytest.approx(float(g["eta_fit_incoh"]), rel=1e-6)

    s = golden("sim_sspec.npz")
    dyn = s["dyn"].astype(np.float64)
    dyn = dyn - np.nanmean(s["dyn"])
    params = [dyn, s["freqs"], s["times"], s["sw_etas"], s["sw_edges"], None, False, float(s["sw_fw"]),
              int(s["sw_npad"]), True, float(s["sw_tau_mask"]), False]
    res = thth.single_search(params)
    np.testing.assert_allclose(res[4], s["sw_eigs"], rtol=1e-6)     # reference ran on float32 input
    assert float(res[0]) == pytest.approx(float(s["sw_eta_fit"]), rel=1e-5)


def test_sspec_vs_reference_golden(golden):
    from scintools_amd.dynspec import Dynspec

    class Obj:
        pass
    g = golden("sim_sspec.npz")
    o = Obj()
    o.dyn, o.freqs, o.times, o.dt, o.df = g["dyn"], g["freqs"], g["times"], float(g["dt"]), float(g["df"])
    d = Dynspec(dyn=o, verbose=False)
    cases = {"default": {}, "prewhite": dict(prewhite=True), "full": dict(halve=False),
             "hamming": dict(window="hamming", window_frac=0.25),
             "blackman_pw": dict(window="blackman", window_frac=0.3, prewhite=True),
             "bartlett": dict(window="bartlett", window_frac=0.2), "nowindow": dict(window=None)}
    for tag, kw in cases.items():
        fdop, tdel, sec = d.calc_sspec(return_sspec=True, **kw)
        ref = g[f"sec64_{tag}"]      # the reference run on a float64 copy of the input
        assert np.array_equal(fdop, g[f"fdop_{tag}"]) and np.array_equal(tdel, g[f"tdel_{tag}"])
        assert sec.shape == ref.shape
        # compare in linear power relative to the spectrum's peak: bins at the rounding
        # floor (1e-16 of the peak and below) have meaningless dB values in both
        lin, lref = 10 ** (sec / 10), 10 ** (ref / 10)
        assert np.abs(lin - lref).max() <= 1e-10 * lref.max(), tag
        strong = lref > 1e-6 * lref.max()
        assert np.abs(sec - ref)[strong].max() <= 1e-8, tag
    fdop, tdel, sec = d.calc_sspec(input_dyn=g["dyn"][:75, :101], prewhite=True)
    lin, lref = 10 ** (sec / 10), 10 ** (g["sub_sec64"] / 10)
    assert np.abs(lin - lref).max() <= 1e-10 * lref.max()
    d.calc_sspec()
    assert d.sspec.shape == g["sec_default"].shape and d.fdop.shape == g["fdop_default"].shape
    # against the reference's own single-precision result the agreement is float32-level
    lin, lref = 10 ** (d.sspec / 10), 10 ** (g["sec_default"] / 10)
    assert np.abs(lin - lref).max() <= 1e-5 * lref.max()


@pytest.mark.parametrize("nf,nt", [(48, 5000), (33, 8192), (128, 9001)])
def test_sspec_long_rows_vs_oracle(nf, nt):
    """Rows longer than 4096 samples pad to 16384+ points: the real-to-complex path that sends
    PAIRS of rows through the decimated row transform (fft.hip, PairRows / PairSplitSource)."""
    import torch
    from oracle import sspec_oracle as so
    from scintools_amd.dynspec import sspec_device
    from scintools_amd.device import to_device
    rng = np.random.default_rng(nf + nt)
    dyn = rng.standard_normal((nf, nt)) + 3.0 + np.cos(np.arange(nt) / 37.0)[None, :]
    for kw in (dict(), dict(prewhite=True), dict(halve=False), dict(window=None)):
        sec = sspec_device(to_device(dyn, torch.float64), **kw).cpu().numpy()
        ref = so.calc_sspec(dyn, 30.0, 0.1, **kw)[2]
        assert sec.shape == ref.shape
        lin, lref = 10 ** (sec / 10), 10 ** (ref / 10)
        assert np.abs(lin - lref).max() <= 1e-10 * lref.max(), kw
        strong = lref > 1e-6 * lref.max()
        assert np.abs(sec - ref)[strong].max() <= 1e-8, kw


@pytest.mark.parametrize("nf,nt,npad", [(32, 16384, 0), (16, 8192, 1), (64, 4096, 3), (31, 16384, 0)])
def test_conjugate_spectrum_long_rows(thth, to, nf, nt, npad):
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, _ = arc_dynspec(nf, nt, seed=nf + npad, nimg=8)
    ref = to.conjugate_spectrum(dyn, npad)
    got = thth.conjugate_spectrum(dyn, npad).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_fit_thetatheta_vs_reference_golden(golden):
    """The Dynspec entry point of the tutorial (dynspec_thth.rst:146-170): 16 chunks of 64
    channels, 52 curvatures each, npad=3, auto-sized edges -- against the reference's own
    fit_thetatheta run (tests/golden/make_golden.py::gen_fit_thetatheta)."""
    from scintools_amd.dynspec import Dynspec
    g = golden("fit_thetatheta.npz")

    class B:
        dyn, freqs, times, dt, df = g["dspec"], g["freq"], g["time"], float(g["dt"]), float(g["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50)
    with pytest.warns(UserWarning, match="does not draw"):
        assert d.thetatheta_single(cf=0, ct=0) is None          # reference defaults: plot=True, arrays=False
    etas, eigs, popt = d.thetatheta_single(cf=0, ct=0, plot=False, arrays=True)
    assert np.array_equal(etas, g["single_etas"])
    np.testing.assert_allclose(eigs, g["single_eigs"], rtol=1e-9)
    np.testing.assert_allclose(popt, g["single_popt"], rtol=1e-6)
    d.fit_thetatheta()
    assert np.array_equal(d.f0s, g["f0s"])
    np.testing.assert_allclose(d.eta_evo, g["eta_evo"], rtol=1e-6)
    np.testing.assert_allclose(d.eta_evo_err, g["eta_evo_err"], rtol=1e-4)
    assert d.ththeta == pytest.approx(float(g["ththeta"]), rel=1e-6)
    assert d.ththetaerr == pytest.approx(float(g["ththetaerr"]), rel=1e-4)
    assert abs(d.ththeta - 44.0 * (1332.0 + 64.0) ** 0 ) < 10     # same arc as the known answer


def test_fit_thetatheta_256_channel_chunk_vs_reference_golden(golden):
    """thetatheta_single / fit_thetatheta beyond the tutorial's 64-channel chunks: ONE 256-channel
    chunk of the same data (npad = 3: CS 1024 x 600; 1166 default edges, N up to 1165, 52 curvatures) against the reference's
    own run (tests/golden/fit_thetatheta_256.npz, make_golden.py fit256: 76 + 83 s of the reference)."""
    from scintools_amd.dynspec import Dynspec
    g, f = golden("fit_thetatheta_256.npz"), golden("fit_thetatheta.npz")
    n = int(g["nchan"])

    class B:
        dyn, freqs, times, dt, df = f["dspec"][:n], f["freq"][:n], f["time"], float(f["dt"]), float(f["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=256, edges_lim=.3, eta_min=30, eta_max=50)
    np.testing.assert_allclose(d.edges, g["edges"], rtol=1e-13)
    assert d.neta == int(g["neta"]) and d.cwt == int(g["cwt"])
    etas, eigs, popt = d.thetatheta_single(cf=0, ct=0, plot=False, arrays=True)
    np.testing.assert_allclose(etas, g["single_etas"], rtol=1e-14)
    np.testing.assert_allclose(eigs, g["single_eigs"], rtol=1e-9)
    np.testing.assert_allclose(popt, g["single_popt"], rtol=1e-6)
    d.fit_thetatheta()
    np.testing.assert_allclose(d.eta_evo, g["eta_evo"], rtol=1e-6)
    np.testing.assert_allclose(d.eta_evo_err, g["eta_evo_err"], rtol=1e-4)
    assert d.ththeta == pytest.approx(float(g["ththeta"]), rel=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_eigvec_and_chisq_sweep_vs_reference_golden(thth, golden, tag):
    """The batched modeler sweep (eigenpairs of all etas in one call, then per-eta back-map,
    inverse FFT, chi^2) against the reference's modeler / chisq_calc goldens."""
    g = golden("thth_small.npz")
    CS, tau, fd, etas, edges = g["CS"], g["tau"], g["fd"], g["etas"], g[f"edges_{tag}"]
    w, V, info = thth.eigvec_sweep(CS, tau, fd, etas, edges)
    assert np.all(info["status"] == 0)
    V = V.cpu().numpy()
    for k in range(3):
        n = int(info["N"][k])
        assert w[k] == pytest.approx(float(g[f"mod_w_{tag}{k}"]), rel=1e-9)
        Vr = g[f"mod_V_{tag}{k}"]
        assert n == Vr.shape[0]
        assert abs(np.linalg.norm(V[k, :n]) - 1) <= 1e-12 and not np.any(V[k, n:])
        assert 1 - abs(np.vdot(Vr, V[k, :n])) <= 1e-9
    chis = thth.chisq_sweep(g["dyn"], CS, tau, fd, etas, edges, 1.0)
    for k in range(3):
        assert chis[k] == pytest.approx(float(g[f"chisq_{tag}{k}"]), rel=1e-9)
    single = [thth.chisq_calc(g["dyn"], CS, tau, fd, e, edges, 1.0) for e in etas]
    np.testing.assert_allclose(chis, single, rtol=1e-9)


def test_eval_sweep_multi_equals_per_chunk_sweeps(thth, to):
    """One batched call over several conjugate spectra == the per-chunk calls, bit for bit."""
    import torch
    from scintools_amd.device import empty
    from scintools_amd.synth import arc_dynspec
    stack = empty((3, 64, 128), torch.complex128)
    grids, etas_list, singles = [], [], []
    for k in range(3):
        dyn, freqs, times, eta_true = arc_dynspec(64, 128, seed=30 + k, nimg=10, f0=1300.0 + 40 * k)
        dyn -= dyn.mean()
        tau, fd = to.fft_axis(freqs, 1.0, 0), to.fft_axis(times, 1000.0, 0)
        thth.conjugate_spectrum(dyn, 0, tau, 0.0, True, out=stack[k])
        edges = np.linspace(-fd.max() / 2, fd.max() / 2, 48) * (1.0 + 0.01 * k)
        etas = np.geomspace(0.5, 2.0, 5 + 2 * k) * eta_true
        grids.append((tau, fd, edges)); etas_list.append(etas)
        singles.append(thth.eval_sweep(stack[k], tau, fd, etas, edges))
    multi = thth.eval_sweep_multi(stack, grids, etas_list, batch=4)
    for a, b in zip(multi, singles):
        assert np.array_equal(a, b)


def test_device_crop_tables_equal_the_host_ones(thth):
    """scint_sweep_keep (the eigenvalue sweep's crop tables, built on the device) against the NumPy expression of
    thth_redmap's crop (ththmod.py:153-155) at the headline shape: same indices, same counts."""
    import bench
    dyn, freqs, times, fd, tau, edges, etas, eta_true = bench.make_workload(512, 256, 4096, seed=3)
    grid = thth._Grid(tau, fd, edges)
    etas = np.concatenate((etas, np.geomspace(1e-4, 1e6, 99) * eta_true))
    keep_t, n_dev = thth._sweep_inputs_dev(grid, etas)
    keep_h, n_h = thth._sweep_inputs(grid, etas)
    assert np.array_equal(n_dev, n_h) and n_h.max() > 2000 and n_h.min() < 10
    kd = keep_t.cpu().numpy()
    for i, n in enumerate(n_h):
        assert np.array_equal(kd[i, :n], keep_h[i, :n])


def _align(a, ref):
    return a * np.exp(-1j * np.angle(np.vdot(ref, a)))


def test_phase_retrieval_vs_reference_golden(thth, golden):
    """thetatheta_chunks (all chunks' eigenpairs in one batched sweep) -> mosaic -> Gerchberg-Saxton on the GPU
    against the reference's run (tests/golden/make_golden.py::gen_retrieval); wavefields compared up to a
    global phase; the pool form (chunk by chunk through single_chunk_retrieval) gives the same chunks."""
    from scintools_amd.dynspec import Dynspec
    g = golden("retrieval.npz")
    f = golden("fit_thetatheta.npz")
    n = int(g["nchan"])

    class B:
        dyn, freqs, times, dt, df = f["dspec"][:n], f["freq"][:n], f["time"], float(f["dt"]), float(f["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50, nedge=128)
    assert np.array_equal(d.edges, g["edges"]) and d.neta == int(g["neta"])
    d.calc_wavefield()
    # SURVEY 8c's 1e-9 for V / rev_map / model holds for the whole retrieval chain: measured on an MI355X
    # (tools/retrieval_deviation.py, round 3) ththeta 1.8e-10, eta_evo 5.0e-10 (a curve_fit on the eigenvalue curves sits
    # in between), chunks 4.5e-11, mosaicked wavefield 4.7e-11, after two Gerchberg-Saxton rounds 1.3e-10
    assert d.ththeta == pytest.approx(float(g["ththeta"]), rel=1e-8)
    np.testing.assert_allclose(d.eta_evo, g["eta_evo"], rtol=1e-8)
    assert d.chunks.shape == (7, 1, 64, 150)
    for cf, key in ((0, "chunk0"), (3, "chunk3")):
        ref = g[key]
        assert np.abs(_align(d.chunks[cf, 0], ref) - ref).max() <= 1e-9 * np.abs(ref).max()
    ref = g["wavefield"]
    assert d.wavefield.shape == ref.shape
    assert np.abs(_align(d.wavefield, ref) - ref).max() <= 1e-9 * np.abs(ref).max()
    d.gerchberg_saxton(niter=2)
    ref = g["wavefield_gs"]
    assert np.abs(_align(d.wavefield, ref) - ref).max() <= 1e-9 * np.abs(ref).max()
    # the reference's idiom, pool.map(single_chunk_retrieval, pars): same chunks up to each chunk's own phase
    class SerialPool:
        def map(self, fn, it):
            return [fn(x) for x in it]
    batched = d.chunks.copy()
    d.thetatheta_chunks(pool=SerialPool())
    for a, b in zip(batched[:, 0], d.chunks[:, 0]):
        assert np.abs(_align(a, b) - b).max() <= 1e-9 * np.abs(b).max()
    # the constraints GS enforces: measured amplitudes, and causality after the last projection
    pos = B.dyn[: ref.shape[0]] > 0
    np.testing.assert_allclose(np.abs(d.wavefield[pos]) ** 2, B.dyn[: ref.shape[0]][pos], rtol=1e-9)


def test_retrieval_tail_on_the_device_equals_the_host_fed_path_bit_for_bit(thth, golden):
    """Round 6: Dynspec.thetatheta_chunks cuts its chunks on the device (ththmod.chunk_cut_device: window, nanmean, nan_to_num and
    the padding value in NumPy's summation order) and calc_wavefield mosaics them there (ththmod.mosaic_device).  Against the
    host-fed forms on the same data -- chunk_retrieval_batch with the reference's three lines per chunk on the host
    (dynspec.py:1782-1790), and the host loop ththmod.mosaic (ththmod.py:1492-1554) -- not a bit differs; a dynamic spectrum with
    NaNs included."""
    from scintools_amd.dynspec import Dynspec
    f = golden("fit_thetatheta.npz")
    n = 256
    dyn = np.array(f["dspec"][:n], dtype=float)
    dyn[17, 40:44] = np.nan

    class B:
        pass
    B.dyn, B.freqs, B.times, B.dt, B.df = dyn, f["freq"][:n], f["time"], float(f["dt"]), float(f["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50, nedge=128)
    d.calc_wavefield()
    chunks_dev, wf_dev = d.chunks.copy(), d.wavefield.copy()
    pars = []
    for cf in range(d.ncf_ret):
        fs = slice(cf * (d.cwf // 2), cf * (d.cwf // 2) + d.cwf)
        freq2 = np.copy(d.freqs[fs])
        eta = d.ththeta * (d.fref / freq2.mean()) ** 2
        for ct in range(d.nct_ret):
            ts = slice(ct * (d.cwt // 2), ct * (d.cwt // 2) + d.cwt)
            dspec2 = np.copy(d.dyn[fs, ts])
            dspec2 -= np.nanmean(dspec2)
            dspec2 = np.nan_to_num(dspec2)
            pars.append((dspec2, d.edges * (freq2.mean() / d.fref), np.copy(d.times[ts]), freq2, eta))
    host_fed = thth.chunk_retrieval_batch(pars, d.npad, d.thth_tau_mask).reshape(chunks_dev.shape)
    assert np.abs(chunks_dev).max() > 0 and np.array_equal(chunks_dev, host_fed)
    assert thth._numpy_mosaic_modes(d.cwf, d.cwt) is not None
    assert np.array_equal(wf_dev, thth.mosaic(chunks_dev))


@pytest.mark.parametrize("shape", [(3, 3, 128, 128), (2, 3, 256, 256), (2, 2, 64, 150)])
def test_device_mosaic_is_the_host_loop_bit_for_bit(thth, shape):
    """ththmod.mosaic_device on the GPU against the host loop ththmod.mosaic (ththmod.py:1492-1554) on random chunks: shapes
    below and above NumPy's temporary-elision threshold (256 KiB: 128 x 128 complex128), several 8192-element buffer pieces."""
    import torch
    from scintools_amd.device import require_gpu
    rng = np.random.default_rng(sum(shape))
    ch = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 10.0 ** rng.integers(-2, 3, shape[:2] + (1, 1))
    got = thth.mosaic_device(torch.from_numpy(ch).to(require_gpu())).cpu().numpy()
    assert np.array_equal(got, thth.mosaic(ch))


def test_ifft2_shifted_and_gs_kernels_vs_numpy(thth):
    import torch
    from scintools_amd.ththmod import _ifft2_shifted_dev, gerchberg_saxton_device
    from oracle import thth_oracle as to
    rng = np.random.default_rng(8)
    for shape in ((64, 32), (96, 150), (33, 17)):
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        ref = np.fft.ifft2(np.fft.ifftshift(x))[: shape[0] // 2, : shape[1] - 3] * 2.5
        got = _ifft2_shifted_dev(thth.to_device(x), scale=2.5, crop=(shape[0] // 2, shape[1] - 3)).cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        dyn = np.abs(rng.standard_normal(shape)) + 0.1
        dyn[3, 5] = -1.0
        dyn[7, 2] = np.nan
        tau = np.fft.fftshift(np.fft.fftfreq(shape[0], 0.1))
        ref = to.gerchberg_saxton(x, dyn, tau, niter=3)
        got = gerchberg_saxton_device(x, dyn, tau, niter=3)
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


def test_calc_acf_vs_oracle_and_reference_golden(golden):
    """BASELINE config 1's other half: Dynspec.calc_acf (direct method)."""
    from oracle import sspec_oracle as so
    from scintools_amd.dynspec import Dynspec
    g = golden("sim_sspec.npz")

    class O:
        dyn, freqs, times, dt, df = g["dyn"].astype(np.float64), g["freqs"], g["times"], float(g["dt"]), float(g["df"])
    d = Dynspec(dyn=O(), verbose=False)
    d.calc_acf()
    ref64 = so.calc_acf(O.dyn)
    assert d.acf.shape == ref64.shape == (192, 256)
    assert np.abs(d.acf - ref64).max() <= 1e-12
    assert np.abs(d.acf - g["acf"]).max() <= 1e-5          # the reference ran on its float32 dyn
    raw = d.calc_acf(input_dyn=O.dyn[:75, :101], normalise=False)   # odd sizes, no mean subtraction
    x = np.fft.fft2(O.dyn[:75, :101], s=[150, 202])
    ref = np.real(np.fft.fftshift(np.fft.ifft2(np.abs(x) ** 2)))
    assert np.abs(raw - ref).max() <= 1e-12 * np.abs(ref).max()


def test_calc_asymmetry_vs_reference_golden(golden):
    from scintools_amd.dynspec import Dynspec
    g = golden("retrieval.npz")
    f = golden("fit_thetatheta.npz")
    n = int(g["nchan"])

    class B:
        dyn, freqs, times, dt, df = f["dspec"][:n], f["freq"][:n], f["time"], float(f["dt"]), float(f["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50, nedge=128)
    d.calc_asymmetry()
    assert d.asymmetry.shape == g["asymmetry"].shape
    np.testing.assert_allclose(d.asymmetry.real, g["asymmetry"].real, rtol=1e-6, atol=1e-8)


def test_diagonal_back_map_fuzz_vs_oracle(thth, to):
    """Round 6, on the hardware (its rsqrt and FMA-free arithmetic, its scheduling): the uniform-grid back-map kernel
    (csrc/thth.hip: rev_diag_kernel -- strided sweeps, per-wavefront segmented scans with carried rows) against the oracle's
    np.histogram2d image on forty random geometries -- odd and shifted axes, theta steps commensurate with the Doppler step
    (s * step ON a column edge), curvatures of either sign from 1 % of the arc's to 20 times it -- to 1e-12 of the peak with
    identical empty-bin masks; the call says which kernel ran; a second call gives the same bits."""
    import torch
    rng = np.random.default_rng(20260930)
    for trial in range(40):
        ntau, nfd = int(rng.integers(16, 2600)), int(rng.integers(8, 200))
        dt, df = 0.0137 * float(rng.uniform(0.5, 2)), 0.211 * float(rng.uniform(0.5, 2))
        tau = (np.arange(ntau) - ntau // 2) * dt
        fd = (np.arange(nfd) - nfd // 2) * df
        kind = trial % 5
        if kind == 1:
            tau, fd = tau + float(rng.uniform(-0.5, 0.5)) * dt, fd + float(rng.uniform(-0.5, 0.5)) * df
        nedge = 2 * int(rng.integers(3, 500))
        lim = float(rng.uniform(0.2, 1.1)) * fd.max() / 2
        if kind == 2:
            lim = df / int(rng.integers(1, 4)) * (nedge - 1) / 2
        if kind == 3:
            lim = df * int(rng.integers(1, 3)) * (nedge - 1) / 2
        edges = np.linspace(-lim, lim, nedge)
        eta = float(10 ** rng.uniform(-2.0, 1.3)) * (1 if rng.uniform() < 0.8 else -1) * np.abs(tau).max() / lim ** 2
        grid = thth._Grid(tau, fd, edges)
        n = grid.M
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        w = np.array([float(rng.uniform(-3, 3))])
        th_t, v_t, w_t = thth.to_device(grid.th_cents, torch.float64), thth.to_device(v), thth.to_device(w, torch.float64)
        info = {}
        rec = thth._rev_map_dev(grid.geom, th_t, n, eta, True, vec_t=v_t, w_t=w_t, info=info).cpu().numpy()
        again = thth._rev_map_dev(grid.geom, th_t, n, eta, True, vec_t=v_t, w_t=w_t).cpu().numpy()
        ref = np.nan_to_num(to.rev_map(np.outer(v, np.conj(v)) * np.abs(w[0]), tau, fd, eta, edges, True))
        key = (trial, ntau, nfd, n, eta)
        assert info["uniform_grid"] == 1, key
        assert np.array_equal(rec, again), key
        assert np.abs(rec - ref).max() <= 1e-12 * np.abs(ref).max(), key
        assert np.array_equal(rec != 0, ref != 0), key


def test_chisq_from_accumulators_fuzz(thth, to, monkeypatch):
    """Round 6, on the hardware: chi^2 from the back-map's accumulators (symmetric axes) against the written-image route
    (SCINT_CHISQ_FUSE=0) to 1e-12 on eight random arcs -- generic theta grids (fused, at most a few curvatures redone) and grids
    whose step is half the Doppler step (every odd diagonal ON a column edge: fused, curvatures redone from a written image) --
    and against the oracle's chisq_calc at two curvatures of each (1e-9)."""
    from scintools_amd.synth import arc_dynspec
    rng = np.random.default_rng(7)
    for trial in range(8):
        nf, nt = 2 * int(rng.integers(100, 700)), 2 * int(rng.integers(24, 100))
        dyn, freqs, times, eta_true = arc_dynspec(nf, nt, seed=100 + trial, nimg=8, noise=0.05)
        dyn = dyn - dyn.mean()
        fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
        CS = to.conjugate_spectrum(dyn, 0)
        if trial % 2:
            n = 2 * int(rng.integers(10, nt // 2))
            edges = (np.arange(n) - (n - 1) / 2) * ((fd[1] - fd[0]) / 2)
        else:
            edges = np.linspace(-fd.max() / 2, fd.max() / 2, 2 * int(rng.integers(20, 150)))
        th = thth._Grid(tau, fd, edges).th_cents
        eta_full = np.abs(tau).max() / (th ** 2).max()
        etas = np.sort(10 ** rng.uniform(-2.0, 0.6, 12)) * eta_full * 1.00071
        monkeypatch.setenv("SCINT_CHISQ_FUSE", "1")
        a, ia = thth.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, return_info=True)
        monkeypatch.setenv("SCINT_CHISQ_FUSE", "0")
        b, ib = thth.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, return_info=True)
        key = (trial, nf, nt, edges.shape[0])
        assert ia["fused"] and not ib["fused"], key
        assert np.all(ia["status"] == 0), key
        if trial % 2:
            assert ia["redone"] >= 1, key
        np.testing.assert_allclose(a, b, rtol=1e-12, err_msg=str(key))
        for i in (2, 9):
            assert a[i] == pytest.approx(to.chisq_calc(dyn, CS, tau, fd, etas[i], edges, 3.0), rel=1e-9), key
