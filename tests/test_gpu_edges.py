"""Edge cases of the domain on the GPU path: empty / 1x1 reduced maps, an all-masked conjugate
spectrum, tiny matrices, non-finite input, dtype/contiguity of inputs, the non-Hermitian
wrap-around, and schedule invariance of the sweep."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import thth_oracle as to
    from scintools_amd import ththmod as thth
    from scintools_amd.device import require_gpu
    from scintools_amd.synth import arc_dynspec
    require_gpu()
    dyn, freqs, times, eta_true = arc_dynspec(128, 128, seed=4, nimg=16)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    CS = to.conjugate_spectrum(dyn, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 96)
    return thth, to, dict(dyn=dyn, fd=fd, tau=tau, CS=CS, edges=edges, eta=eta_true)


def test_crop_to_nothing_gives_nan_like_the_reference(env):
    thth, to, p = env
    # eta so large that theta^2 eta < tau_max keeps only the theta = 0 centre (or nothing)
    etas = np.array([p["eta"], 1e6 * p["eta"], 1e9 * p["eta"]])
    eigs, info = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], return_info=True)
    assert np.isfinite(eigs[0]) and info["status"][0] == 0
    assert info["N"][1] <= 1 and info["N"][2] <= 1
    assert np.isnan(eigs[1]) and np.isnan(eigs[2])          # reference: exception -> nan
    for e in etas[1:]:
        with pytest.raises(Exception):
            to.Eval_calc(p["CS"], p["tau"], p["fd"], e, p["edges"])
    fit = thth.fit_eig_peak(etas, eigs, 0.1)                 # NaNs are dropped before the fit
    assert fit[2] is None or np.isfinite(fit[0])


def test_chisq_sweep_with_curvatures_that_keep_one_two_or_all_centres(env):
    """Lopsided edges whose centres are -1, 0, 0.1, 1 (x fd.max / 4): a curvature that keeps all four gives the oracle's chi^2,
    one that keeps two (no mean edge step: the reference's rev_map raises, ththmod.py:166) or one is NaN, and the sweep goes on."""
    thth, to, p = env
    s = p["fd"].max() / 2
    edges = np.array([-1.95, -0.05, 0.05, 0.15, 1.85]) * s / 2
    th = (edges[1:] + edges[:-1]) / 2
    th = th - th[np.abs(th) == np.abs(th).min()]
    tmax = np.abs(p["tau"]).max()
    etas = np.array([0.5 * tmax / th[-1] ** 2, 0.5 * (tmax / th[2] ** 2 + tmax / th[-1] ** 2), 4.0 * tmax / th[2] ** 2,
                     0.25 * tmax / th[-1] ** 2])
    got = thth.chisq_sweep(p["dyn"], p["CS"], p["tau"], p["fd"], etas, edges, 2.0)
    ref = [to.chisq_calc(p["dyn"], p["CS"], p["tau"], p["fd"], e, edges, 2.0) for e in etas[[0, 3]]]
    np.testing.assert_allclose(got[[0, 3]], ref, rtol=1e-9)
    assert np.isnan(got[1]) and np.isnan(got[2])
    with pytest.raises(Exception):
        to.chisq_calc(p["dyn"], p["CS"], p["tau"], p["fd"], etas[1], edges, 2.0)


def test_all_masked_cs_gives_zero(env):
    thth, to, p = env
    eigs = thth.eval_sweep(np.zeros_like(p["CS"]), p["tau"], p["fd"], np.array([p["eta"]]), p["edges"])
    assert eigs[0] == 0.0
    red, _ = thth.thth_redmap(np.zeros_like(p["CS"]), p["tau"], p["fd"], p["eta"], p["edges"])
    assert not np.any(red)


def test_tiny_reduced_matrices_against_lapack(env):
    thth, to, p = env
    for nedge in (4, 6, 8, 10, 34):
        edges = np.linspace(-p["fd"].max() / 2, p["fd"].max() / 2, nedge)
        red, _ = to.thth_redmap(p["CS"], p["tau"], p["fd"], p["eta"], edges)
        eig, info = thth.eval_sweep(p["CS"], p["tau"], p["fd"], np.array([p["eta"]]), edges, return_info=True)
        assert info["N"][0] == red.shape[0] == nedge - 1
        assert eig[0] == pytest.approx(abs(np.linalg.eigvalsh(red)[-1]), rel=1e-10, abs=1e-9)


def test_inputs_float32_noncontiguous_and_nan(env):
    thth, to, p = env
    # complex64 / Fortran-ordered CS are converted, result as for the converted array
    cs32 = p["CS"].astype(np.complex64)
    a = thth.thth_redmap(np.asfortranarray(cs32), p["tau"], p["fd"], p["eta"], p["edges"])[0]
    b = to.thth_redmap(cs32.astype(np.complex128), p["tau"], p["fd"], p["eta"], p["edges"])[0]
    assert np.array_equal(a, b)
    # NaN pixels in the CS: the Hermitian branch turns them into zeros (nan_to_num, ththmod.py:114)
    cs = p["CS"].copy()
    cs[70, 80] = np.nan + 1j * np.nan
    cs[75, 90] = np.inf
    a = thth.thth_redmap(cs, p["tau"], p["fd"], p["eta"], p["edges"])[0]
    b = to.thth_redmap(cs, p["tau"], p["fd"], p["eta"], p["edges"])[0]
    assert np.all(np.isfinite(a))
    ok = np.isfinite(b.real) & np.isfinite(b.imag) & (np.abs(b) < 1e300) & (np.abs(a) < 1e300)
    assert np.array_equal(a[ok], b[ok])


def test_nonhermitian_wraparound_and_indexerror(env):
    thth, to, p = env
    # edges wider than the CS: negative fd indices wrap like NumPy's fancy indexing ...
    edges = np.linspace(-0.9 * p["fd"].max(), 0.9 * p["fd"].max(), 64)
    ref = to.thth_map(p["CS"], p["tau"], p["fd"], p["eta"], edges, hermetian=False)
    got = thth.thth_map(p["CS"], p["tau"], p["fd"], p["eta"], edges, hermetian=False)
    assert np.array_equal(got, ref)
    # ... and raise once they fall below -len(fd), exactly where NumPy raises
    edges = np.linspace(-1.6 * p["fd"].max(), 1.6 * p["fd"].max(), 64)
    with pytest.raises(IndexError):
        to.thth_map(p["CS"], p["tau"], p["fd"], 0.01 * p["eta"], edges, hermetian=False)
    with pytest.raises(IndexError):
        thth.thth_map(p["CS"], p["tau"], p["fd"], 0.01 * p["eta"], edges, hermetian=False)


def test_sweep_is_schedule_invariant(env):
    """Continuous batching must not change a bit: any batch size, any eta order."""
    thth, to, p = env
    etas = np.geomspace(0.3, 3.0, 23) * p["eta"]
    ref = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], batch=23)
    for b in (1, 2, 7):
        assert np.array_equal(thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], batch=b), ref)
    perm = np.random.default_rng(0).permutation(len(etas))
    got = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas[perm], p["edges"], batch=5)
    assert np.array_equal(got, ref[perm])
    one = np.array([thth.Eval_calc(p["CS"], p["tau"], p["fd"], e, p["edges"]) for e in etas[:4]])
    assert np.array_equal(one, ref[:4])
    # by default two groups of slots run on two streams and the host queues two chunks ahead of the
    # convergence flags; one group and/or synchronous scheduling (depth 1) must give the same
    # bits, for eigenvalues and for eigenvectors (the check cadence, `every`, is the one switch that may move a value
    # inside its tolerance)
    from scintools_amd import _lib
    lib = _lib.load()
    w2, V2, _ = thth.eigvec_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], batch=6)
    for depth, every, groups in ((1, 0, 2), (2, 0, 1), (1, 0, 1), (2, 3, 2), (1, 1, 1)):
        assert lib.scint_sweep_schedule(depth, every, groups) == 0
        try:
            e1 = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], batch=6)
            w1, V1, _ = thth.eigvec_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], batch=6)
        finally:
            assert lib.scint_sweep_schedule(0, 0, 0) == 0
        if every == 0:
            assert np.array_equal(e1, ref)
            assert np.array_equal(w1, w2) and np.array_equal(V1.cpu().numpy(), V2.cpu().numpy())
        else:
            # another check cadence looks at the stopping rule after other passes: a curvature may stop a pass or two
            # earlier or later -- another Ritz value of the same matrix under the same bound, not another bit pattern
            # of the same one (first GPU run of round 4: 98088.27666578 against ...577)
            np.testing.assert_allclose(e1, ref, rtol=4 * thth.DEFAULT_TOL)
            np.testing.assert_allclose(w1, w2, rtol=4 * thth.DEFAULT_TOL)
    assert lib.scint_sweep_schedule(3, 0, 0) != 0 and lib.scint_sweep_schedule(0, 17, 0) != 0      # out of range: refused


def test_dynspec_with_nans_goes_through_fit_thetatheta(env):
    from scintools_amd.dynspec import Dynspec
    thth, to, p = env
    dyn = p["dyn"] + 50.0
    dyn[5, 7] = np.nan
    dyn[40:42, 100] = np.nan

    class B:
        pass
    b = B()
    b.dyn, b.freqs, b.times = dyn, 1400.0 + 0.05 * np.arange(128), 30.0 * np.arange(128)
    d = Dynspec(dyn=b, verbose=False)
    d.prep_thetatheta(cwf=64, cwt=128, eta_min=0.5 * p["eta"], eta_max=2 * p["eta"], nedge=64, npad=1, fw=0.3)
    d.fit_thetatheta()
    assert d.eta_evo.shape == (2, 1) and np.all(np.isfinite(d.thth_eigs))


def test_non_default_stream_and_iteration_cap(env):
    import torch
    thth, to, p = env
    etas = np.geomspace(0.5, 2.0, 6) * p["eta"]
    ref = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        cs = thth.conjugate_spectrum(p["dyn"], 0, pad_value=0.0)
        got = thth.eval_sweep(cs, p["tau"], p["fd"], etas, p["edges"])
        red = thth.thth_redmap(cs, p["tau"], p["fd"], etas[2], p["edges"])[0]
    s.synchronize()
    np.testing.assert_allclose(got, ref, rtol=1e-12)
    assert np.array_equal(red, to.thth_redmap(cs.cpu().numpy(), p["tau"], p["fd"], etas[2], p["edges"])[0])
    # an iteration cap that cannot converge -> status NOCONV -> NaN (the reference's nan on failure)
    capped, info = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], max_iter=6, return_info=True)
    assert np.all(np.isnan(capped)) and np.all(info["status"] == 4) and np.all(info["iters"] == 6)


def test_threads_with_their_own_streams(env):
    """Two Python threads, each on its own stream and workspace, give the single-thread answer."""
    import threading
    import torch
    thth, to, p = env
    etas = np.geomspace(0.5, 2.0, 9) * p["eta"]
    ref = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"])
    out = {}

    def work(k):
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(3):
                out[k] = thth.eval_sweep(p["CS"], p["tau"], p["fd"], etas, p["edges"], batch=2 + k)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert np.array_equal(out[0], ref) and np.array_equal(out[1], ref)


def test_nearly_rank_one_matrix_vs_arpack(env):
    """A noiseless screen with one dominant image: theta-theta is close to rank one, the start
    vector (row n/2) is already nearly the eigenvector, and beta_j << |alpha_j| from the first
    Lanczos steps on -- the regime where beta_j^2 = |u_j|^2 - alpha_j^2 cancels.  The eigenvalue
    must still match ARPACK to the parity bar, for eigenvalue and eigenvector sweeps."""
    from scintools_amd.synth import arc_dynspec
    thth, to, p = env
    dyn, freqs, times, eta_true = arc_dynspec(256, 256, seed=9, nimg=6, noise=0.0)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 256)
    CS = to.conjugate_spectrum(dyn, 0)
    etas = np.array([0.9, 1.0, 1.1]) * eta_true
    ref = np.array([to.Eval_calc(CS, tau, fd, e, edges) for e in etas])
    eigs, info = thth.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
    assert np.all(info["status"] == 0)
    np.testing.assert_allclose(eigs, ref, rtol=1e-9)
    w, V, vinfo = thth.eigvec_sweep(CS, tau, fd, etas, edges)
    np.testing.assert_allclose(np.abs(w), ref, rtol=1e-9)
    red, _ = to.thth_redmap(CS, tau, fd, etas[1], edges)
    v = V[1, : red.shape[0]].cpu().numpy()
    assert np.linalg.norm(red @ v - w[1] * v) <= 1e-8 * abs(w[1])


def test_rank_deficient_theta_theta_against_lapack(env):
    """Screens of two or three images give a theta-theta whose weight sits in a handful of eigenvectors: the block
    Krylov space is numerically exhausted after a few steps and the Cholesky pivots of the two-vector recurrence
    turn into rounding noise (relative pivot floor).  Largest algebraic eigenvalue against LAPACK on
    the oracle's gathered matrix, across the curvature range and for an iteration cap far beyond the rank."""
    thth, to, p = env
    from scintools_amd.synth import arc_dynspec
    for nimg, seed in ((1, 3), (2, 5), (3, 9)):
        dyn, freqs, times, eta_true = arc_dynspec(128, 128, seed=seed, nimg=nimg)
        dyn -= dyn.mean()
        fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
        CS = to.conjugate_spectrum(dyn, 0)
        edges = np.linspace(-fd.max() / 2, fd.max() / 2, 96)
        etas = np.array([0.6, 1.0, 1.7]) * eta_true
        eigs, info = thth.eval_sweep(CS, tau, fd, etas, edges, max_iter=400, return_info=True)
        assert np.all(info["status"] == 0)
        for e, got in zip(etas, eigs):
            lam = np.linalg.eigvalsh(to.thth_redmap(CS, tau, fd, e, edges)[0])
            assert got == pytest.approx(abs(lam[-1]), rel=1e-9)


def test_secondary_spectrum_zero_subnormal_and_nonfinite_powers():
    """calc_sspec of a constant dynamic spectrum (every power exactly 0 -> -inf), of one with a NaN sample (NaN everywhere) and of
    one whose powers are subnormal: the branch-free logarithm of sspec_rows2_kernel hands such values to the series form, and the
    result has -inf / NaN where NumPy has them (dynspec.py:3685-3721)."""
    import warnings
    import torch
    from oracle import sspec_oracle
    from scintools_amd.device import require_gpu
    from scintools_amd.dynspec import sspec_device
    dev = require_gpu()
    rng = np.random.default_rng(1)
    nan_in = 1.0 + rng.standard_normal((300, 260))
    nan_in[3, 177] = np.nan
    for dyn in (np.full((300, 260), 2.5), nan_in, 1e-160 * rng.standard_normal((300, 260))):
        sec = sspec_device(torch.from_numpy(dyn).to(dev)).cpu().numpy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = sspec_oracle.calc_sspec(dyn, 30.0, 1.0)[2]
        assert np.array_equal(np.isneginf(sec), np.isneginf(ref)) and np.array_equal(np.isnan(sec), np.isnan(ref))
        fin = np.isfinite(ref)
        if fin.any():
            assert np.abs(sec - ref)[fin].max() <= 1e-8
