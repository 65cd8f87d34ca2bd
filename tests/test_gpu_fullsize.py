"""Full-size (BASELINE.json configs 2-5) checks on the GPU.  Where the CPU oracle finishes in
seconds the comparison is direct; otherwise size-independent properties are used: Parseval and
Hermitian symmetry of the conjugate spectrum, exact Hermitian symmetry / zero diagonal of
theta-theta, the eigen-residual |A v - w v| of the returned pair (checked with an independent
NumPy mat-vec), agreement between the packed sweep and the dense solver, linearity."""
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
    return thth, to, arc_dynspec


def _setup(env, size, seed, nedge=None):
    thth, to, arc = env
    dyn, freqs, times, eta_true = arc(size, size, seed=seed, nimg=64)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, nedge or size)
    return dyn, fd, tau, edges, eta_true


def test_cs_4096_parseval_symmetry_linearity(env):
    thth, to, _ = env
    dyn, fd, tau, edges, _ = _setup(env, 4096, 3)
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0).cpu().numpy()
    assert cs.shape == (4096, 4096)
    # Parseval
    assert np.sum(np.abs(cs) ** 2) == pytest.approx(dyn.size * np.sum(dyn**2), rel=1e-12)
    # real input: CS[-tau, -fd] = conj(CS[tau, fd]) (fftshifted: index n - i, for i >= 1)
    a = cs[1:, 1:]
    assert np.abs(a - np.conj(a[::-1, ::-1])).max() <= 1e-9 * np.abs(cs).max()
    # a few rows against NumPy directly (full 2-D transform of 4096^2 is cheap enough on the host)
    ref = np.fft.fftshift(np.fft.fft2(dyn))
    assert np.abs(cs - ref).max() <= 1e-12 * np.abs(ref).max()
    # linearity
    cs2 = thth.conjugate_spectrum(2.5 * dyn, 0, pad_value=0.0).cpu().numpy()
    assert np.abs(cs2 - 2.5 * cs).max() <= 1e-12 * np.abs(cs2).max()


def test_sspec_4096_vs_oracle(env):
    import torch
    from oracle import sspec_oracle as so
    from scintools_amd.dynspec import sspec_device
    thth, to, _ = env
    dyn, *_ = _setup(env, 4096, 3)
    for kw in (dict(), dict(prewhite=True)):
        sec = sspec_device(thth.to_device(dyn, torch.float64), **kw).cpu().numpy()
        ref = so.calc_sspec(dyn, 30.0, 0.1, **kw)[2]
        assert sec.shape == ref.shape == (4096, 8192)
        lin, lref = 10 ** (sec / 10), 10 ** (ref / 10)
        assert np.abs(lin - lref).max() <= 1e-10 * lref.max()
        strong = lref > 1e-6 * lref.max()
        assert np.abs(sec - ref)[strong].max() <= 1e-8


@pytest.mark.timeout(900)
def test_config3_with_npad3_eigenvalue_vs_oracle(env):
    """BASELINE config 3 with the reference's default padding npad = 3: conjugate spectrum 16384^2 (4 GiB on
    the GPU), the gather reading the 4x finer grid.  One curvature against the oracle (NumPy fft2 of the
    padded 16384^2 plane + gather + ARPACK: about a minute of host time), rtol 1e-9."""
    thth, to, arc = env
    size = 4096
    dyn, freqs, times, eta_true = arc(size, size, seed=3, nimg=64)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 3), to.fft_axis(freqs, 1.0, 3)
    assert fd.shape[0] == 4 * size and tau.shape[0] == 4 * size
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    eta = 0.84 * eta_true
    cs = thth.conjugate_spectrum(dyn, 3)
    assert tuple(cs.shape) == (4 * size, 4 * size)
    got, info = thth.eval_sweep(cs, tau, fd, np.array([eta]), edges, return_info=True)
    assert info["status"][0] == 0
    del cs
    CS = to.conjugate_spectrum(dyn, 3)
    ref = to.Eval_calc(CS, tau, fd, eta, edges)
    assert got[0] == pytest.approx(ref, rel=1e-9)


@pytest.mark.timeout(900)
def test_headline_sweep_16_etas_vs_oracle(env):
    """The headline workload itself (bench.py's default: 4096^2, nedge 4096, 256 eta over geomspace(0.25, 4) eta_true):
    the whole 256-point curve from ONE sweep call, then 16 of its curvatures spread over the sweep -- cropped sizes
    N from 2447 to 4095, both flat ends and the peak -- against the oracle's Eval_calc (NumPy gather + ARPACK,
    ththmod.py:371-401), rtol 1e-9.  (bench.py reports the same comparison in `cpu_baseline.max_rel_diff_vs_gpu`;
    VERDICT r3 weak 1b asked for it as a test.)"""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _setup(env, 4096, 3)
    etas = np.geomspace(0.25, 4.0, 256) * eta_true
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    eigs, info = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
    assert np.all(info["status"] == 0) and (int(info["N"].min()), int(info["N"].max())) == (2447, 4095)
    CS = cs.cpu().numpy()
    idx = [int(round(k)) for k in np.linspace(0, 255, 16)]
    assert len(set(int(info["N"][i]) for i in idx)) >= 5            # the sample spans the crop
    for i in idx:
        ref = to.Eval_calc(CS, tau, fd, etas[i], edges)
        assert eigs[i] == pytest.approx(ref, rel=1e-9), (i, int(info["N"][i]))
    assert abs(etas[int(np.argmax(eigs))] / eta_true - 1) < 0.02    # and the curve peaks at the injected curvature


@pytest.mark.parametrize("size,seed", [(2048, 2), (4096, 3)])
def test_thth_hermitian_and_eigpair_residual(env, size, seed):
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _setup(env, size, seed)
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    eta = 0.84 * eta_true
    red, edges_red = thth.thth_redmap(cs, tau, fd, eta, edges)
    n = red.shape[0]
    assert n == size - 1
    assert np.array_equal(red, red.conj().T)                 # exactly Hermitian
    assert not np.any(np.diag(red)) and not np.any(np.diag(red[::-1]))
    assert np.count_nonzero(red) > 0.5 * n * n
    # dense solver: eigenpair residual with an independent mat-vec
    from scintools_amd.ththmod import _eigh_top_dev
    w, V, iters = _eigh_top_dev(thth.to_device(red))
    V = V.cpu().numpy()
    assert np.linalg.norm(red @ V - w * V) <= 1e-9 * abs(w)
    assert abs(np.linalg.norm(V) - 1) <= 1e-12
    # packed sweep agrees with the dense solver and with Eval_calc's definition |w|
    eig = thth.eval_sweep(cs, tau, fd, np.array([eta]), edges)[0]
    assert eig == pytest.approx(abs(w), rel=1e-10)
    # Rayleigh quotient can only underestimate the top eigenvalue
    x = np.random.default_rng(0).standard_normal(n) + 0j
    assert (np.vdot(x, red @ x).real / np.vdot(x, x).real) <= w * (1 + 1e-12)


def test_gather_4096_spot_rows_vs_oracle(env):
    """Bit-equality of whole rows of the 4095 x 4095 map against the oracle's index math,
    evaluated row-wise on the host (the full oracle map needs ~1 GB of temporaries per array)."""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _setup(env, 4096, 3)
    CS = np.fft.fftshift(np.fft.fft2(dyn))
    red, _ = thth.thth_redmap(CS, tau, fd, eta_true, edges)
    th = to.theta_centres(edges)
    dtau, dfd = np.diff(tau).mean(), np.diff(fd).mean()
    M = th.shape[0]
    for i in (0, 1, 777, 2047, 3000, 4093):
        th1, th2 = th, th[i]
        tau_inv = (((eta_true * (th1**2 - th2**2)) - tau[0] + dtau / 2) // dtau).astype(int)
        fd_inv = (((th1 - th2) - fd[0] + dfd / 2) // dfd).astype(int)
        pnts = (tau_inv > 0) * (tau_inv < tau.shape[0]) * (fd_inv < fd.shape[0])
        row = np.zeros(M, complex)
        row[pnts] = CS[tau_inv[pnts], fd_inv[pnts]]
        row *= np.sqrt(np.abs(2 * eta_true * (th2 - th1)))
        row[: i + 1] = 0                       # strictly upper part of row i
        row[M - 1 - i] = 0                     # anti-diagonal
        got = red[i].copy()
        got[: i + 1] = 0
        assert np.array_equal(got, np.nan_to_num(row)), i


def test_config5_8192_eigen_and_fit_vs_oracle(env):
    """BASELINE config 5: 8192^2, N = 8191, fp64 eigenvalue tolerance-checked against the
    reference algorithm (oracle gather + ARPACK) at the curvature of the arc, and the fitted
    curvature of a 33-eta sweep around it against the injected one."""
    thth, to, _ = env
    dyn, fd, tau, edges, eta_true = _setup(env, 8192, 5)
    cs = thth.conjugate_spectrum(dyn, 0, pad_value=0.0)
    etas = np.linspace(0.9, 1.1, 33) * eta_true
    eigs, info = thth.eval_sweep(cs, tau, fd, etas, edges, return_info=True)
    assert np.all(info["status"] == 0) and info["N"].max() == 8191
    eta_fit, eta_sig, _ = thth.fit_eig_peak(etas, eigs, 0.1)
    assert eta_fit == pytest.approx(eta_true, rel=2e-2)
    CS = cs.cpu().numpy()
    ref = to.Eval_calc(CS, tau, fd, etas[16], edges)
    assert eigs[16] == pytest.approx(ref, rel=1e-9)
    # SURVEY 8c's tolerance for the FITTED curvature is against the reference algorithm, not against the injected value:
    # |d eta| <= 1e-6 eta.  Nine curvatures of the sweep through the oracle (gather + ARPACK at N = 8191), the same
    # parabola fit (ththmod.py:814-859) on both nine-point curves.
    sub = np.arange(0, 33, 4)
    ref9 = np.array([ref if i == 16 else to.Eval_calc(CS, tau, fd, etas[i], edges) for i in sub])
    np.testing.assert_allclose(eigs[sub], ref9, rtol=1e-9)
    fit_gpu, sig_gpu, _ = thth.fit_eig_peak(etas[sub], eigs[sub], 0.1)
    fit_ref, sig_ref, _ = to.fit_eig_peak(etas[sub], ref9, 0.1)
    assert np.isfinite(fit_ref) and abs(fit_gpu - fit_ref) <= 1e-6 * abs(fit_ref)
    assert abs(fit_gpu - fit_ref) < 1e-3 * sig_ref
