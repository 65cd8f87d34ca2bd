"""The mixed-precision eigenvalue sweep on the GPU (scint_sweep_precision(1); csrc/eigen_packed.hip, "Mixed precision"):
the Lanczos passes stream a complex64 copy of theta-theta, the value returned is the Ritz value of a certificate pass on
the complex128 tiles under the float64 sweep's own a-posteriori bound.  Parity bar: the one of the float64 sweep (rtol 1e-9
against the reference's ARPACK values), and agreement with the float64 sweep far inside it -- both return Ritz values of the
same float64 matrix.  (The file name sorts last: the float64 library is tested first.)"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import sim_oracle, thth_oracle
    from scintools_amd import ththmod as thth
    from scintools_amd.device import require_gpu
    require_gpu()
    return thth, thth_oracle, sim_oracle


@pytest.fixture()
def mixed(env):
    thth = env[0]
    assert thth.sweep_precision("mixed") == "f64"
    yield thth
    assert thth.sweep_precision("f64") == "mixed"


def _stats():
    from scintools_amd import _lib
    st = (ctypes.c_double * 4)()
    _lib.check(_lib.load().scint_sweep_stats(st), "scint_sweep_stats")
    return dict(bytes32=st[0], bytes64=st[1], certified=st[2], cert_passes=st[3])


def _arc(to, size, seed=3, nimg=64):
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=seed, nimg=nimg)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    return dyn, fd, tau, edges, eta_true


def test_reference_simulation_screen_every_curvature(env, golden):
    """The reference's own Eval_calc curve on its Simulation screen at 1024^2 (tests/golden/sim_sweep.npz, 96 curvatures,
    flat parts with lambda_2 / lambda_1 -> 0.99 included): rtol 1e-9 for every curvature, and the float64 sweep's values
    to 1e-12."""
    thth, to, so = env
    g = golden("sim_sweep.npz")
    sim = so.baseline_dynspec(1024, int(g["s1024_seed"]))
    assert so.checksum(sim.dyn) == str(g["s1024_sha256"])
    dyn = np.array(sim.dyn, dtype=np.float64)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(sim.times, 1000.0, 0), to.fft_axis(sim.freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 1024)
    etas, ref = g["s1024_etas"], g["s1024_eigs"]
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    e64, i64 = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
    thth.sweep_precision("mixed")
    try:
        emx, imx = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
        st = _stats()
    finally:
        thth.sweep_precision("f64")
    assert np.all(imx["status"] == 0)
    rel = np.abs(emx - ref) / np.abs(ref)
    assert rel.max() <= 1e-9, (int(np.argmax(rel)), float(rel.max()))
    np.testing.assert_allclose(emx, e64, rtol=1e-12)
    assert st["certified"] == len(etas)
    assert st["cert_passes"] <= 2.0 * len(etas)              # as a rule ONE complex128 pass per curvature (a failed first step costs two more)
    assert imx["iters"].mean() <= i64["iters"].mean() + 3    # the iteration phase costs what the float64 sweep costs


def test_2048_sweep_against_the_float64_sweep(env):
    """Config-2 size, 48 curvatures over the sweep's range: values of the float64 sweep to 1e-12, one certificate each,
    bytes by operand as counted by the library (4 n (n + 1) per complex64 pass, 8 n (n + 1) per complex128 pass)."""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _arc(to, 2048)
    etas = np.geomspace(0.25, 4.0, 48) * eta_true
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    e64, i64 = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
    thth.sweep_precision("mixed")
    try:
        emx, imx = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
        st = _stats()
        again = thth.eval_sweep(cs, tau, fd, etas, edges, batch=5)
    finally:
        thth.sweep_precision("f64")
    assert np.all(imx["status"] == 0) and np.all(i64["status"] == 0)
    np.testing.assert_allclose(emx, e64, rtol=1e-12)
    assert np.array_equal(emx, again)                        # batch size, slot grouping, arrival order: same bits
    n_ = imx["N"].astype(float)
    assert st["certified"] == len(etas)
    cert = st["cert_passes"]
    assert st["bytes64"] >= np.sum(8 * n_ * (n_ + 1)) and cert >= len(etas)
    if cert == len(etas):
        assert st["bytes64"] == np.sum(8 * n_ * (n_ + 1))
        assert st["bytes32"] == np.sum(4 * n_ * (n_ + 1) * (imx["iters"] - 1))


def test_headline_size_spot_check(env):
    """4096^2 (N = 4095: 64 block rows, strips of 14 tiles x 4 rows), four curvatures against the float64 sweep."""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _arc(to, 4096)
    etas = np.array([0.3, 0.84, 1.0, 2.5]) * eta_true
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    e64 = thth.eval_sweep(cs, tau, fd, etas, edges)
    thth.sweep_precision("mixed")
    try:
        emx, imx = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
    finally:
        thth.sweep_precision("f64")
    assert np.all(imx["status"] == 0)
    np.testing.assert_allclose(emx, e64, rtol=1e-12)


def test_units_of_the_data_and_edge_cases(mixed, env):
    """Data scaled by 2^-200 / 2^+150 (outside the float32 range): the same bits, scaled.  All-zero spectrum, crop to
    nothing, non-finite input, iteration cap, tiny matrices: what the float64 sweep does."""
    thth, to, _ = env
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(128, 128, seed=4, nimg=16)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    CS = to.conjugate_spectrum(dyn, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 96)
    etas = np.array([0.7, 1.0, 1.4]) * eta_true
    base, info = mixed.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
    assert np.all(info["status"] == 0)
    ref = np.array([to.Eval_calc(CS, tau, fd, e, edges) for e in etas])
    np.testing.assert_allclose(base, ref, rtol=1e-9)
    for k in (-200, 150):
        assert np.array_equal(mixed.eval_sweep(CS * 2.0 ** k, tau, fd, etas, edges), base * 2.0 ** k)
    assert mixed.eval_sweep(np.zeros_like(CS), tau, fd, etas[:1], edges)[0] == 0.0
    eigs, info = mixed.eval_sweep(CS, tau, fd, np.array([eta_true, 1e9 * eta_true]), edges, return_info=True)
    assert np.isfinite(eigs[0]) and np.isnan(eigs[1]) and info["status"][1] == 5
    bad = CS.copy()
    bad[bad.shape[0] // 2 + 3, bad.shape[1] // 2 + 5] = np.nan
    got, info = mixed.eval_sweep(bad, tau, fd, etas[1:2], edges, return_info=True)
    mixed.sweep_precision("f64")
    want, i64 = mixed.eval_sweep(bad, tau, fd, etas[1:2], edges, return_info=True)
    mixed.sweep_precision("mixed")
    assert info["status"][0] == i64["status"][0] and np.array_equal(np.isnan(got), np.isnan(want))
    eigs, info = mixed.eval_sweep(CS, tau, fd, etas[1:2], edges, max_iter=3, return_info=True)
    assert info["status"][0] == 4 and np.isnan(eigs[0])
    for nedge in (4, 6, 10, 34):
        e2 = np.linspace(-fd.max() / 2, fd.max() / 2, nedge)
        red, _ = to.thth_redmap(CS, tau, fd, eta_true, e2)
        eig, info = mixed.eval_sweep(CS, tau, fd, np.array([eta_true]), e2, return_info=True)
        assert info["N"][0] == red.shape[0] and info["status"][0] == 0
        assert eig[0] == pytest.approx(abs(np.linalg.eigvalsh(red)[-1]), rel=1e-10, abs=1e-9)


def test_certificate_measures_the_float64_residual(mixed, env):
    """The certificate has to see the residual the complex64 copy leaves (~1e-8 |theta|, eight digits below the vectors it
    is formed from: pk2_cert_resid_kernel computes it row by row, the Gram form would lose it).  Asked for more than that
    residual allows (tol = 1e-17), no certificate may pass at its first step; the runs continue on the complex128 tiles and
    end with the float64 sweep's values."""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _arc(to, 512, seed=5, nimg=24)
    etas = np.geomspace(0.5, 2.0, 12) * eta_true
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    mixed.sweep_precision("f64")
    ref, iref = mixed.eval_sweep(cs, tau, fd, etas, edges, tol=1e-17, return_info=True)
    mixed.sweep_precision("mixed")
    got, info = mixed.eval_sweep(cs, tau, fd, etas, edges, tol=1e-17, return_info=True)
    st = _stats()
    assert np.all(info["status"] == 0) and np.all(iref["status"] == 0)
    np.testing.assert_allclose(got, ref, rtol=1e-13)
    assert st["certified"] == len(etas) and st["cert_passes"] >= 2 * len(etas)
    got, info = mixed.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
    assert _stats()["cert_passes"] <= 2.0 * len(etas)


def test_single_search_and_fit_thetatheta_vs_the_reference(mixed, golden):
    """The callers of the sweep in mixed mode against the reference's OWN runs (tests/golden, unmodified reference): the
    chunk search on the tutorial data -- eigenvalue curve rtol 1e-9, fitted curvature 1e-6 -- and Dynspec.fit_thetatheta
    (every chunk's sweep in one batched call) with the tolerances of tests/test_gpu_parity.py."""
    g = golden("thth_sample.npz")
    params = [g["chunk"], g["freq"], g["time"], g["etas"], g["edges"], None, False, 0.1,
              int(g["npad"]), True, 0.0, False]
    eta_fit, eta_sig, fm, tm, eigs = mixed.single_search(params)
    np.testing.assert_allclose(eigs, g["eigs"], rtol=1e-9)
    assert float(eta_fit) == pytest.approx(float(g["eta_fit"]), rel=1e-6)
    assert float(eta_sig) == pytest.approx(float(g["eta_sig"]), rel=1e-4)
    assert _stats()["certified"] == len(g["etas"])
    from scintools_amd.dynspec import Dynspec
    f = golden("fit_thetatheta.npz")

    class B:
        dyn, freqs, times, dt, df = f["dspec"], f["freq"], f["time"], float(f["dt"]), float(f["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50)
    etas, eigs, popt = d.thetatheta_single(cf=0, ct=0, plot=False, arrays=True)
    np.testing.assert_allclose(eigs, f["single_eigs"], rtol=1e-9)
    d.fit_thetatheta()
    np.testing.assert_allclose(d.eta_evo, f["eta_evo"], rtol=1e-6)
    np.testing.assert_allclose(d.eta_evo_err, f["eta_evo_err"], rtol=1e-4)
    assert d.ththeta == pytest.approx(float(f["ththeta"]), rel=1e-6)


def test_mixed_all_eigenpair_and_chisq_sweeps_against_the_float64_ones(env):
    """scint_sweep_precision(2), "mixed-all" (round 4): the eigenPAIR sweeps iterate on the complex64 copy to the eigenvalue
    rule and finish the vector on the complex128 tiles, to the float64 sweep's own residual rule.  2048^2, twelve curvatures
    over the sweep's range: eigenvalues of the float64 eigenpair sweep to 1e-12, eigenvectors to 1e-9 of their largest entry
    after removing the one global phase, chi^2 (ththmod.py:330-368) to the parity bar of the float64 path, 1e-9 -- and one
    curvature of it against the oracle's chisq_calc; fewer bytes streamed than by the float64 eigenpair sweep."""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _arc(to, 2048)
    etas = np.geomspace(0.3, 3.5, 12) * eta_true
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    w64, v64, i64 = thth.eigvec_sweep(cs, tau, fd, etas, edges)
    b64 = _stats()
    c64 = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, 1.0)
    assert thth.sweep_precision("mixed-all") == "f64"
    try:
        wmx, vmx, imx = thth.eigvec_sweep(cs, tau, fd, etas, edges)
        bmx = _stats()
        cmx = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, 1.0)
        cmx2 = thth.chisq_sweep(dyn, cs, tau, fd, etas, edges, 1.0)
    finally:
        assert thth.sweep_precision("f64") == "mixed-all"
    assert np.all(imx["status"] == 0) and np.all(i64["status"] == 0)
    np.testing.assert_allclose(wmx, w64, rtol=1e-12)
    a_, b_ = vmx.cpu().numpy(), v64.cpu().numpy()
    for a, b in zip(a_, b_):
        ph = np.vdot(b, a) / abs(np.vdot(b, a))
        assert np.abs(a / ph - b).max() <= 1e-9 * np.abs(b).max()
    np.testing.assert_allclose(cmx, c64, rtol=1e-9)
    assert np.array_equal(cmx, cmx2)                              # bit-reproducible from call to call
    assert bmx["bytes32"] > 0 and bmx["certified"] == len(etas)
    assert bmx["bytes64"] + bmx["bytes32"] < b64["bytes64"]
    k = 5
    ref = to.chisq_calc(dyn, to.conjugate_spectrum(dyn, 0), tau, fd, etas[k], edges, 1.0)
    assert abs(cmx[k] - ref) <= 1e-9 * abs(ref)
