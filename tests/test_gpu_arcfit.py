"""GPU parity for the arc-normalisation row: scale_dyn('lambda'),
calc_sspec(lamsteps), norm_sspec, fit_arc and the fit_arc fallback of prep_thetatheta, against
the reference's own outputs (tests/golden/arcfit.npz) and the CPU oracle on fresh inputs.

Tolerances:
  lamdyn (cubic spline)   1e-12 of the array maximum (tridiagonal moments vs scipy's banded B-spline solve)
  normsspec 2-D           bit-equal, NaNs included (np.interp branch structure and rounding reproduced)
  mask                    identical
  powerspectrum / avg     rtol 1e-12 (summation order; 10**x from the device libm)
  fit_arc eta, errors     rtol 1e-9
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_oracle_golden import _NORM_CASES  # noqa: E402  (same cases as the oracle pin)


@pytest.fixture(scope="module")
def ds():
    from scintools_amd import dynspec
    from scintools_amd.device import require_gpu
    require_gpu()
    return dynspec


class _Obj:
    pass


def _dynspec(ds, g, key="dyn"):
    o = _Obj()
    o.dyn, o.freqs, o.times = np.array(g[key]), np.array(g["freqs"]), np.array(g["times"])
    o.dt, o.df, o.freq = float(g["dt"]), float(g["df"]), float(g["freq"])
    return ds.Dynspec(dyn=o, verbose=False)


def _filled(a):
    return np.ma.filled(np.ma.array(a, dtype=float), np.nan)


def test_scale_dyn_lambda_vs_reference(ds, golden):
    g = golden("arcfit.npz")
    d = _dynspec(ds, g)
    d.scale_dyn()
    assert d.lamdyn.shape == g["lamdyn"].shape
    assert np.array_equal(d.lam, g["lam"]) and d.dlam == float(g["dlam"]) and d.nlam == len(g["lam"])
    err = np.abs(d.lamdyn - g["lamdyn"]).max() / np.abs(g["lamdyn"]).max()
    assert err < 1e-12, err
    d.calc_sspec(lamsteps=True)
    assert np.array_equal(d.beta, g["beta"]) and np.array_equal(d.fdop, g["fdop"])
    assert np.array_equal(d.tdel, g["tdel"])
    big = g["lamsspec"] > g["lamsspec"].max() - 150
    assert np.abs(d.lamsspec - g["lamsspec"])[big].max() < 1e-8


@pytest.mark.parametrize("order", ["descending", "shuffled"])
def test_scale_dyn_lambda_axis_orders(ds, golden, order):
    """interp1d sorts its axis: a descending or shuffled channel order gives the same lamdyn."""
    g = golden("arcfit.npz")
    d = _dynspec(ds, g)
    perm = np.arange(len(d.freqs))[::-1] if order == "descending" else np.random.default_rng(4).permutation(len(d.freqs))
    d.freqs, d.dyn = d.freqs[perm], d.dyn[perm]
    d.scale_dyn()
    assert np.abs(d.lamdyn - g["lamdyn"]).max() / np.abs(g["lamdyn"]).max() < 1e-12


def _with_golden_spectra(ds, g):
    d = _dynspec(ds, g)
    d.lamsspec, d.sspec = np.array(g["lamsspec"]), np.array(g["sspec"])
    d.beta, d.tdel, d.fdop = np.array(g["beta"]), np.array(g["tdel"]), np.array(g["fdop"])
    return d


@pytest.mark.parametrize("tag", sorted(_NORM_CASES))
def test_norm_sspec_vs_reference(ds, golden, tag):
    g = golden("arcfit.npz")
    d = _with_golden_spectra(ds, g)
    kw = dict(_NORM_CASES[tag])
    eta = kw.pop("eta", float(g["fa_betaeta"]))
    frac = kw.pop("delmax_frac", None)
    if frac is not None:
        kw["delmax"] = frac * np.max(g["tdel"])
    d.norm_sspec(eta=eta, plot=False, **kw)
    assert np.array_equal(d.normsspec_fdop, g[f"{tag}_fdop"])
    assert np.array_equal(d.normsspec_tdel, g[f"{tag}_tdel"])
    assert np.array_equal(np.ma.getmaskarray(d.normsspec), g[f"{tag}_mask"])
    assert np.array_equal(d.mask, g[f"{tag}_mask"])
    got, ref = np.asarray(d.normsspec.data), g[f"{tag}_norm"]
    if tag == "nc":      # subtract_artefacts: the delay response is a device nanmean (summation order)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11, equal_nan=True)
    else:
        assert np.array_equal(got, ref, equal_nan=True)
    np.testing.assert_allclose(_filled(d.powerspectrum), g[f"{tag}_pow"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(_filled(d.weights), g[f"{tag}_weights"], rtol=1e-11, equal_nan=True)
    np.testing.assert_allclose(_filled(d.normsspecavg), g[f"{tag}_avg"], rtol=1e-11, atol=1e-11, equal_nan=True)


def test_fit_arc_vs_reference(ds, golden):
    g = golden("arcfit.npz")
    d = _with_golden_spectra(ds, g)
    d.fit_arc(lamsteps=True, numsteps=2000)
    assert d.noise == pytest.approx(float(g["fa_noise"]), rel=1e-12)
    np.testing.assert_allclose(d.eta_array, g["fa_eta_array"], rtol=1e-14)
    np.testing.assert_allclose(d.norm_sspec_avg, g["fa_spec"], rtol=1e-11)
    np.testing.assert_allclose(d.prob_eta_peak, g["fa_prob"], rtol=1e-8)
    assert d.betaeta == pytest.approx(float(g["fa_betaeta"]), rel=1e-9)
    assert d.betaetaerr == pytest.approx(float(g["fa_betaetaerr"]), rel=1e-9)
    assert d.betaetaerr2 == pytest.approx(float(g["fa_betaetaerr2"]), rel=1e-6)
    d.fit_arc(lamsteps=True, numsteps=1500, asymm=True, log_parabola=True, logsteps=True, weighted=True,
              etamin=40.0, etamax=4000.0, constraint=[100, 2000], nsmooth=7, startbin=4, cutmid=5,
              delmax=0.8 * np.max(g["tdel"]))
    assert d.betaeta_left == pytest.approx(float(g["fb_left"]), rel=1e-9)
    assert d.betaeta_right == pytest.approx(float(g["fb_right"]), rel=1e-9)
    assert d.betaetaerr_left == pytest.approx(float(g["fb_lefterr"]), rel=1e-9)
    assert d.betaetaerr_right == pytest.approx(float(g["fb_righterr"]), rel=1e-9)
    np.testing.assert_allclose(d.norm_sspec_avg1, g["fb_spec1"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(d.norm_sspec_avg2, g["fb_spec2"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(d.eta_array, g["fb_eta_array"], rtol=1e-14)


def test_fit_arc_end_to_end_from_dynspec(ds, golden):
    """Whole chain on the device: spline resample -> secondary spectrum -> normalisation -> fit."""
    g = golden("arcfit.npz")
    d = _dynspec(ds, g)
    d.fit_arc(lamsteps=True, numsteps=2000)
    assert d.betaeta == pytest.approx(float(g["fa_betaeta"]), rel=1e-7)
    assert d.betaetaerr == pytest.approx(float(g["fa_betaetaerr"]), rel=1e-7)
    assert d.noise == pytest.approx(float(g["fa_noise"]), rel=1e-9)
    with pytest.raises(TypeError):          # the reference divides the default list, dynspec.py:1147
        d.fit_arc(lamsteps=False, numsteps=2000)


def test_prep_thetatheta_bounds_from_fit_arc(ds, golden):
    g, f = golden("arcfit.npz"), golden("fit_thetatheta.npz")
    o = _Obj()
    o.dyn, o.freqs, o.times, o.dt, o.df = f["dspec"], f["freq"], f["time"], float(f["dt"]), float(f["df"])
    d = ds.Dynspec(dyn=o, verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3)
    assert d.betaeta == pytest.approx(float(g["pt_betaeta"]), rel=1e-7)
    assert d.eta_min == pytest.approx(float(g["pt_eta_min"]), rel=1e-7)
    assert d.eta_max == pytest.approx(float(g["pt_eta_max"]), rel=1e-7)
    assert d.neta == int(g["pt_neta"])
    np.testing.assert_allclose(d.edges, g["pt_edges"], rtol=1e-12)


@pytest.mark.parametrize("shape,seed", [((300, 200), 1), ((512, 768), 2), ((1024, 1024), 3)])
def test_norm_sspec_vs_oracle_fresh_inputs(ds, shape, seed):
    """Seeded arcs the goldens do not cover: odd shapes, non-power-of-two axes, every option."""
    from oracle import arcfit_oracle as ao
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, _ = arc_dynspec(shape[0], shape[1], seed=seed, nimg=32)
    o = _Obj()
    o.dyn, o.freqs, o.times = dyn, freqs, times
    d = ds.Dynspec(dyn=o, verbose=False)
    d.calc_sspec(lamsteps=True)
    ref = ao.calc_sspec_lam(dyn, freqs, d.dt, d.df)
    assert np.abs(d.lamdyn - ref["lamdyn"]).max() / np.abs(ref["lamdyn"]).max() < 1e-12
    # identical spectrum into both sides, so that the normalisation itself is compared bit for bit
    sspec = np.array(d.lamsspec)
    eta = d.beta[len(d.beta) // 2] / (0.5 * d.fdop.max())**2
    for kw in (dict(), dict(logsteps=True, numsteps=777, cutmid=3, startbin=2),
               dict(maxnormfac=2.5, minnormfac=0.1, weighted=False, powerspec_cut=True),
               dict(subtract_artefacts=True, delmax=0.6 * d.tdel.max(), cutmid=6)):
        d.norm_sspec(eta=eta, lamsteps=True, **kw)
        r = ao.norm_sspec(sspec, d.beta, d.tdel, d.fdop, d.freq, eta, lamsteps=True, **kw)
        assert np.array_equal(d.mask, np.ma.getmaskarray(r["normsspec"]))
        if kw.get("subtract_artefacts"):
            np.testing.assert_allclose(np.asarray(d.normsspec.data), np.asarray(r["normsspec"].data), rtol=0,
                                       atol=1e-10, equal_nan=True)
        else:
            assert np.array_equal(np.asarray(d.normsspec.data), np.asarray(r["normsspec"].data), equal_nan=True)
        np.testing.assert_allclose(_filled(d.powerspectrum), _filled(r["powerspectrum"]), rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(_filled(d.normsspecavg), _filled(r["normsspecavg"]), rtol=1e-10, atol=1e-10,
                                   equal_nan=True)


def test_norm_sspec_errors(ds, golden):
    g = golden("arcfit.npz")
    d = _with_golden_spectra(ds, g)
    d.fdop = d.fdop + 0.3 * (d.fdop[1] - d.fdop[0])       # an axis without an exact zero
    with pytest.raises(ValueError, match="array of sample points is empty"):
        d.norm_sspec(eta=1e12, lamsteps=True)             # no Doppler bin inside the first row's arc
    d.fdop = np.array(g["fdop"])
    for kw in (dict(plot=True), dict(velocity=True), dict(interp_nan=True), dict(fit_spectrum=True)):
        with pytest.raises(NotImplementedError):
            d.norm_sspec(eta=float(g["fa_betaeta"]), **kw)


# ------------------------------------------------------------------ kernels, directly
def _interp_rows(ds, sspec, fdop, yaxis, eta, maxnormfac, x, cut=(0, 0), offset=None, xlin=None, row0=0):
    import torch
    from scintools_amd import _lib
    from scintools_amd.device import empty, ptr, stream_ptr, to_device
    lib = _lib.load()
    nrow, nc = sspec.shape
    nr, nx = nrow - row0, len(x)
    t = lambda a: None if a is None else to_device(np.ascontiguousarray(a, dtype=float), torch.float64)
    s_t, f_t, y_t, x_t, o_t, l_t = t(sspec), t(fdop), t(yaxis), t(x), t(offset), t(xlin)
    norm, mask, pw = empty((nr, nx), torch.float64), empty((nr, nx), torch.uint8), empty((nr,), torch.float64)
    rc = lib.scint_norm_sspec(ptr(s_t), nc, nc, ptr(f_t), ptr(y_t), row0, nr, float(eta), float(maxnormfac),
                              int(cut[0]), int(cut[1]), ptr(o_t), ptr(x_t), ptr(l_t), nx, ptr(norm), ptr(mask),
                              ptr(pw), stream_ptr())
    _lib.check(rc, "scint_norm_sspec")
    return norm.cpu().numpy(), mask.cpu().numpy().astype(bool), pw.cpu().numpy()


def _interp_rows_numpy(sspec, fdop, yaxis, eta, maxnormfac, x, cut=(0, 0), offset=None, row0=0):
    s = np.array(sspec[row0:], dtype=float)
    s[:, cut[0]:cut[1]] = np.nan
    if offset is not None:
        s = s - offset[:, None]
    out, mask = [], []
    for r in range(s.shape[0]):
        scale = np.sqrt(yaxis[row0 + r] / eta)
        sel = abs(fdop) <= maxnormfac * scale
        xp = fdop[sel] / scale
        out.append(np.interp(x, xp, s[r, sel]))
        mask.append((np.abs(x) > np.max(np.abs(xp))) | np.isnan(out[-1]))
    return np.array(out), np.array(mask)


@pytest.mark.parametrize("axis", ["uniform", "irregular", "coarse"])
def test_interp_kernel_bit_exact_incl_nonfinite(ds, axis):
    """np.interp reproduced bit for bit on axes that defeat the uniform-grid guess, with NaN,
    +-inf and repeated values in the data, x exactly on sample points and outside the range."""
    rng = np.random.default_rng(5)
    nc, nrow = (257, 40) if axis != "coarse" else (9, 12)
    if axis == "irregular":
        fdop = np.sort(np.concatenate([rng.uniform(-20, 20, nc - 60), rng.normal(0, 0.05, 60)]))
    else:
        fdop = np.linspace(-20, 20, nc)
    sspec = rng.standard_normal((nrow, nc)) * 10 + 30
    sspec[3, 100 % nc] = np.nan
    sspec[4, 5 % nc] = -np.inf
    sspec[5, 7 % nc] = np.inf
    sspec[6, :] = 2.5                       # equal neighbours: the dy[j] == dy[j+1] retry branch
    sspec[6, nc // 3] = np.nan
    sspec[7, nc // 2:] = -np.inf
    yaxis = np.linspace(0.1, 4.0, nrow)
    eta = 0.01
    x = np.concatenate([np.linspace(-3, 3, 801), fdop[::7] / np.sqrt(yaxis[nrow // 2] / eta), [np.nan, -50.0, 50.0]])
    for cut, off in (((0, 0), None), ((nc // 2 - 2, nc // 2 + 2), rng.standard_normal(nrow - 1))):
        got, gmask, pw = _interp_rows(ds, sspec, fdop, yaxis, eta, 3.0, x, cut=cut, offset=off, row0=1)
        ref, rmask = _interp_rows_numpy(sspec, fdop, yaxis, eta, 3.0, x, cut=cut, offset=off, row0=1)
        assert np.array_equal(got, ref, equal_nan=True)
        assert np.array_equal(gmask, rmask)
        lin = np.where(rmask | ~np.isfinite(ref), np.nan, 10**(ref / 10))
        with np.errstate(invalid="ignore"):
            want = np.array([np.nanmean(r) if np.any(~np.isnan(r)) else np.nan for r in lin])
        np.testing.assert_allclose(pw, want, rtol=1e-12, equal_nan=True)


def test_masked_colavg_rownanmean_blockstd(ds):
    import ctypes
    import torch
    from scintools_amd import _lib
    from scintools_amd.device import empty, ptr, stream_ptr, to_device
    lib = _lib.load()
    rng = np.random.default_rng(8)
    nr, nx = 131, 300
    a = rng.standard_normal((nr, nx)) * 5 + 20
    m = rng.random((nr, nx)) < 0.3
    m[:, 17] = True                                     # a column with no entry
    w = rng.uniform(0.5, 2.0, nr)
    sel = (rng.random(nr) < 0.7).astype(np.uint8)
    a_t, m_t = to_device(a, torch.float64), to_device(m.astype(np.uint8), torch.uint8)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_masked_colavg_workspace_bytes(nr, nx, ctypes.byref(need)))
    ws = empty((need.value,), torch.uint8)
    for rowsel in (None, sel):
        avg, none = empty((nx,), torch.float64), empty((nx,), torch.uint8)
        r_t = None if rowsel is None else to_device(rowsel, torch.uint8)
        _lib.check(lib.scint_masked_colavg(ptr(a_t), ptr(m_t), nr, nx, ptr(to_device(w, torch.float64)), ptr(r_t),
                                           ptr(avg), ptr(none), ptr(ws), ws.numel(), stream_ptr()))
        rows = np.ones(nr, bool) if rowsel is None else rowsel.astype(bool)
        ref = np.ma.average(np.ma.array(a, mask=m)[rows], axis=0, weights=w[rows])
        assert np.array_equal(none.cpu().numpy().astype(bool), np.ma.getmaskarray(ref))
        got = avg.cpu().numpy()
        ok = ~np.ma.getmaskarray(ref)
        np.testing.assert_allclose(got[ok], np.asarray(ref[ok]), rtol=1e-13)
        assert np.all(got[~ok] == 0.0)
    # nanmean over selected columns of a row block
    b = a.copy()
    b[rng.random(b.shape) < 0.1] = np.nan
    colsel = (rng.random(nx) < 0.4).astype(np.uint8)
    out = empty((nr - 3,), torch.float64)
    _lib.check(lib.scint_row_nanmean(ptr(to_device(b, torch.float64)), nx, nx, 3, nr - 3,
                                     ptr(to_device(colsel, torch.uint8)), 40, 44, ptr(out), stream_ptr()))
    bb = b[3:].copy()
    bb[:, 40:44] = np.nan
    np.testing.assert_allclose(out.cpu().numpy(), np.nanmean(bb[:, colsel.astype(bool)], axis=1), rtol=1e-13)
    # std of the outer block
    std = empty((1,), torch.float64)
    w2 = empty((8 * 1032,), torch.uint8)
    _lib.check(lib.scint_block_std(ptr(a_t), nx, nx, nr // 2, nr, 140, 155, ptr(std), ptr(w2), w2.numel(), stream_ptr()))
    ref = np.std(np.concatenate((a[nr // 2:, 155:].ravel(), a[nr // 2:, :140].ravel())))
    assert float(std.cpu().numpy()[0]) == pytest.approx(ref, rel=1e-13)


@pytest.mark.parametrize("nf", [4, 5, 33, 700, 1500])
def test_spline_resample_irregular_knots(ds, nf):
    """Not-a-knot cubic spline on irregular channel spacing, down to scipy's 4-point minimum."""
    import torch
    from scipy.interpolate import interp1d
    from scintools_amd.arcfit import spline_resample_device
    from scintools_amd.device import to_device
    rng = np.random.default_rng(nf)
    x = np.sort(rng.uniform(1000, 1400, nf))
    y = rng.standard_normal((nf, 70)) * 3 + 7
    f = np.concatenate([[x[0], x[-1]], rng.uniform(x[0], x[-1], 41), x[1:-1]])
    got = spline_resample_device(to_device(y, torch.float64), x, f).cpu().numpy()
    ref = np.flipud(np.stack([interp1d(x, y[:, k], kind="cubic")(f) for k in range(y.shape[1])], axis=1))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
    with pytest.raises(ValueError):
        spline_resample_device(to_device(y, torch.float64), x, np.array([x[0] - 1.0]))


def test_spline_resample_nonfinite_pixel_poisons_its_whole_column(ds):
    """nf large enough for the frequency-blocked Thomas sweeps: a NaN / inf pixel must turn its
    WHOLE time column NaN, exactly as scipy's interp1d(kind='cubic') (the reference,
    dynspec.py:3948-3957) does, and every other column must be untouched."""
    import torch
    from scipy.interpolate import interp1d
    from scintools_amd.arcfit import spline_resample_device
    from scintools_amd.device import to_device
    nf, nt = 1536, 48
    rng = np.random.default_rng(7)
    x = 1200.0 + 0.25 * np.arange(nf)
    y = rng.standard_normal((nf, nt)) + 3
    f = np.linspace(x[0], x[-1], 900)
    clean = spline_resample_device(to_device(y, torch.float64), x, f).cpu().numpy()
    y[700, 5] = np.nan
    y[20, 17] = np.inf
    got = spline_resample_device(to_device(y, torch.float64), x, f).cpu().numpy()
    ref = np.flipud(np.stack([interp1d(x, y[:, k], kind="cubic")(f) for k in range(nt)], axis=1))
    bad = np.zeros(nt, dtype=bool)
    bad[[5, 17]] = True
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    assert not np.isfinite(got[:, bad]).any()
    assert np.abs(got[:, ~bad] - ref[:, ~bad]).max() / np.abs(ref[:, ~bad]).max() < 1e-12
    # the sequential fallback and the blocked sweep agree to rounding on the clean columns
    assert np.abs(got[:, ~bad] - clean[:, ~bad]).max() <= 1e-12 * np.abs(clean).max()
