"""modeler / rev_map / chisq_calc at the BASELINE sizes (2048^2: config 2, 4096^2: config 3)
against the CPU oracle (ththmod.py:176-368).

At 4096^2 the conjugate spectrum has more delay rows than one workgroup of the rev_map kernel
accumulates, so every back-map takes the multi-slab path (several tau slabs per Doppler column)
that the small goldens never reach; 2048^2 is the largest single-slab case.  One oracle `modeler` per size is computed once (about 8 s at 2048^2 and
about 40 s at 4096^2 of host time) and shared by the tests of that size.

Tolerances: recov / model <= 1e-9 of the array maximum with IDENTICAL empty-bin masks;
w rtol 1e-9; 1 - |<V, V_ref>| <= 1e-9; chi^2 rtol 1e-9.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(2048, 2), (4096, 3)]


@pytest.fixture(scope="module")
def env():
    from oracle import thth_oracle as to
    from scintools_amd import ththmod as thth
    from scintools_amd.device import require_gpu
    require_gpu()
    return thth, to


@pytest.fixture(scope="module", params=SIZES, ids=lambda p: f"{p[0]}sq")
def case(env, request):
    """Workload + oracle modeler + GPU modeler of one size.  Module-scoped and parametrised:
    pytest groups the tests by size, so each oracle model is computed once."""
    size, seed = request.param
    thth, to = env
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=seed, nimg=64)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    eta = 0.93 * eta_true
    CS = to.conjugate_spectrum(dyn, 0)
    ref = to.modeler(CS, tau, fd, eta, edges)
    got = thth.modeler(CS, tau, fd, eta, edges)
    return dict(size=size, dyn=dyn, fd=fd, tau=tau, edges=edges, eta=eta, CS=CS, ref=ref, got=got)


def _assert_image_close(got, ref, rel, tag):
    assert got.shape == ref.shape, tag
    assert np.array_equal(got == 0, ref == 0), f"{tag}: empty-bin masks differ"
    assert np.abs(got - ref).max() <= rel * np.abs(ref).max(), tag


def test_modeler_vs_oracle(env, case):
    """thth_red bit-equal; eigenpair, rank-1 model, back-mapped CS (multi-slab rev_map, rank-1
    path) and model dynamic spectrum within 1e-9 of the oracle's (ththmod.py:274-327)."""
    c = case
    size = c["size"]
    ref, got = c["ref"], c["got"]
    assert got[0].shape == (size - 1, size - 1)
    assert np.array_equal(got[0], ref[0])                                # thth_red
    assert np.array_equal(got[4], ref[4])                                # edges_red
    assert got[5] == pytest.approx(ref[5], rel=1e-9)                     # w
    assert 1 - abs(np.vdot(ref[6], got[6])) <= 1e-9                      # V up to a phase
    assert np.abs(got[1] - ref[1]).max() <= 1e-9 * np.abs(ref[1]).max()  # thth2_red (phase-free)
    _assert_image_close(got[2], ref[2], 1e-9, "recov")
    assert np.abs(got[3] - ref[3]).max() <= 1e-9 * np.abs(ref[3]).max()  # model


def test_rev_map_explicit_hermitian_vs_oracle(env, case):
    """rev_map on an explicit N x N matrix (the reference's own call, ththmod.py:321), Hermitian
    mirror pass included, against the oracle's histogram on the same matrix."""
    thth, to = env
    c = case
    thth2_ref, edges_red = c["ref"][1], c["ref"][4]
    got = thth.rev_map(thth2_ref, c["tau"], c["fd"], c["eta"], edges_red, hermetian=True)
    _assert_image_close(got, c["ref"][2], 1e-9, "rev_map hermitian")


def test_rev_map_not_hermitian_vs_oracle(env, case):
    """The phase-retrieval form (ththmod.py:1459-1463): only the theta_2 = 0 row filled, no mirror
    pass, plus a dense random block so that every slab of every column receives points."""
    thth, to = env
    c = case
    size = c["size"]
    V, w, edges_red = c["ref"][6], c["ref"][5], c["ref"][4]
    n = V.shape[0]
    E = np.zeros((n, n), dtype=complex)
    E[n // 2, :] = np.conjugate(V) * np.sqrt(w)
    rng = np.random.default_rng(size)
    blk = slice(n // 4, n // 4 + 257)
    E[blk, :] += rng.standard_normal((257, n)) + 1j * rng.standard_normal((257, n))
    ref = to.rev_map(E, c["tau"], c["fd"], c["eta"], edges_red, hermetian=False)
    got = thth.rev_map(E, c["tau"], c["fd"], c["eta"], edges_red, hermetian=False)
    _assert_image_close(got, ref, 1e-9, "rev_map non-hermitian")


def test_chisq_calc_vs_oracle(env, case):
    """chisq_calc (ththmod.py:330-368) with the default mask and with an explicit one; the
    oracle value is formed from the oracle model already computed for this size."""
    thth, to = env
    c = case
    dyn, model_ref = c["dyn"], c["ref"][3]
    N = float(dyn.size)
    ref = np.sum((model_ref[: dyn.shape[0], : dyn.shape[1]] - dyn) ** 2) / N
    got = thth.chisq_calc(dyn, c["CS"], c["tau"], c["fd"], c["eta"], c["edges"], N)
    assert got == pytest.approx(ref, rel=1e-9)
    mask = np.ones(dyn.shape, dtype=bool)
    mask[::7, :] = False
    mask[:, 5::11] = False
    ref_m = np.sum((model_ref[: dyn.shape[0], : dyn.shape[1]] - dyn)[mask] ** 2) / N
    got_m = thth.chisq_calc(dyn, c["CS"], c["tau"], c["fd"], c["eta"], c["edges"], N, mask=mask)
    assert got_m == pytest.approx(ref_m, rel=1e-9)


def test_chisq_sweep_vs_chisq_calc(env, case):
    """The batched modeler sweep against per-eta chisq_calc on four curvatures spread over the
    BASELINE range (different reduced sizes N), and against the oracle at the one curvature the
    oracle model exists for."""
    thth, to = env
    c = case
    dyn = c["dyn"]
    N = float(dyn.size)
    eta_true = c["eta"] / 0.93
    etas = np.array([0.3, 0.93, 1.7, 3.6]) * eta_true
    cs_t = thth.to_device(c["CS"])
    chis, info = thth.chisq_sweep(dyn, cs_t, c["tau"], c["fd"], etas, c["edges"], N, return_info=True)
    assert np.all(info["status"] == 0)
    assert len(set(int(v) for v in info["N"])) >= 3            # the crop really changes N
    for e, chi in zip(etas, chis):
        one = thth.chisq_calc(dyn, cs_t, c["tau"], c["fd"], e, c["edges"], N)
        assert chi == pytest.approx(one, rel=1e-9)
    ref = np.sum((c["ref"][3] - dyn) ** 2) / N
    assert chis[1] == pytest.approx(ref, rel=1e-9)
    # the minimum of chi^2 sits at the injected curvature
    fine = np.linspace(0.9, 1.1, 9) * eta_true
    chi_f = thth.chisq_sweep(dyn, cs_t, c["tau"], c["fd"], fine, c["edges"], N)
    assert abs(fine[np.argmin(chi_f)] / eta_true - 1) <= 0.06


def test_chisq_sweep_partner_table_changes_no_bit(env, case):
    """Round 5: the curvatures of a sweep that keep the same theta centres share one partner table for their back-maps
    (launch_rev_walk_table; `crop_group` of scint_chisq_sweep) instead of each walking the centres.  At full size, twelve
    curvatures that keep every centre (one group: a table) and three that crop on their own: chi^2 is the same BITS with the
    table and without it (share_walk=False)."""
    thth, to = env
    c = case
    dyn = c["dyn"]
    eta_true = c["eta"] / 0.93
    etas = np.concatenate([np.linspace(0.3, 1.3, 12), [3.0, 3.4, 3.8]]) * eta_true
    cs_t = thth.to_device(c["CS"])
    a, info = thth.chisq_sweep(dyn, cs_t, c["tau"], c["fd"], etas, c["edges"], float(dyn.size), return_info=True)
    b = thth.chisq_sweep(dyn, cs_t, c["tau"], c["fd"], etas, c["edges"], float(dyn.size), share_walk=False)
    assert np.all(info["status"] == 0) and np.all(np.isfinite(a))
    assert np.sum(info["N"] == info["N"].max()) >= 8 and len(set(int(v) for v in info["N"])) >= 3
    assert np.array_equal(a, b)


@pytest.mark.timeout(900)
def test_cropped_modeler_4096_vs_oracle(env):
    """A curvature whose crop bites (3.5 eta_true at 4096^2: N = 2617 of M = 4095) against the ORACLE's modeler --
    until round 4 the oracle model at 4096^2 existed only at 0.93 eta_true, where N = M, and the cropped curvatures of
    the chi^2 sweep were compared with the product's own per-eta chisq_calc (missing 6).  thth_red and
    edges_red bit-equal; w, V, thth2, recov (identical empty-bin masks), model to 1e-9; chi^2 from chisq_calc and
    from the batched sweep to 1e-9 of the oracle's."""
    thth, to = env
    from scintools_amd.synth import arc_dynspec
    size = 4096
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    eta = 3.5 * eta_true
    CS = to.conjugate_spectrum(dyn, 0)
    ref = to.modeler(CS, tau, fd, eta, edges)
    cs_t = thth.to_device(CS)
    got = thth.modeler(cs_t, tau, fd, eta, edges)
    n = ref[0].shape[0]
    assert n == 2617 and got[0].shape == (n, n)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[4], ref[4])
    assert got[5] == pytest.approx(ref[5], rel=1e-9)
    assert 1 - abs(np.vdot(ref[6], got[6])) <= 1e-9
    assert np.abs(got[1] - ref[1]).max() <= 1e-9 * np.abs(ref[1]).max()
    _assert_image_close(got[2], ref[2], 1e-9, "recov (cropped)")
    assert np.abs(got[3] - ref[3]).max() <= 1e-9 * np.abs(ref[3]).max()
    N = float(dyn.size)
    chi_ref = np.sum((ref[3] - dyn) ** 2) / N
    assert thth.chisq_calc(dyn, cs_t, tau, fd, eta, edges, N) == pytest.approx(chi_ref, rel=1e-9)
    chis, info = thth.chisq_sweep(dyn, cs_t, tau, fd, np.array([0.5, 3.5, 3.9]) * eta_true, edges, N, return_info=True)
    assert int(info["N"][1]) == n and np.all(info["status"] == 0)
    assert chis[1] == pytest.approx(chi_ref, rel=1e-9)


def test_rev_map_and_modeler_are_bit_reproducible(env, case):
    """np.histogram2d is deterministic, so the back-map must be too: the per-pixel sums are
    accumulated on a fixed binary grid (order-independent), and repeated calls -- explicit and
    rank-1, Hermitian and not -- return identical bits."""
    thth, to = env
    c = case
    thth2_ref, edges_red = c["ref"][1], c["ref"][4]
    a = thth.rev_map(thth2_ref, c["tau"], c["fd"], c["eta"], edges_red, hermetian=True)
    for _ in range(2):
        assert np.array_equal(a, thth.rev_map(thth2_ref, c["tau"], c["fd"], c["eta"], edges_red, hermetian=True))
    b = thth.rev_map(thth2_ref, c["tau"], c["fd"], c["eta"], edges_red, hermetian=False)
    assert np.array_equal(b, thth.rev_map(thth2_ref, c["tau"], c["fd"], c["eta"], edges_red, hermetian=False))
    m1 = thth.modeler(c["CS"], c["tau"], c["fd"], c["eta"], c["edges"])
    m2 = thth.modeler(c["CS"], c["tau"], c["fd"], c["eta"], c["edges"])
    assert np.array_equal(m1[2], m2[2]) and np.array_equal(m1[3], m2[3]) and m1[5] == m2[5]
    dyn = c["dyn"]
    x1 = thth.chisq_calc(dyn, c["CS"], c["tau"], c["fd"], c["eta"], c["edges"], 1.0)
    assert x1 == thth.chisq_calc(dyn, c["CS"], c["tau"], c["fd"], c["eta"], c["edges"], 1.0)


def test_chisq_sweep_tail_batches_and_band_ends_vs_oracle(env):
    """the batched tail of the chi^2 sweep (<= 8 curvatures per tail batch, back-map and chi^2 confined to the
    delay band a curvature can reach) directly against the ORACLE's chisq_calc (ththmod.py:330-368) at 4096^2: twelve curvatures
    over geomspace(0.25, 4) eta_true (two tail batches at least), of which the two band ends -- 0.25 eta_true: the narrowest
    delay band, 4 eta_true: the strongest crop -- and two in between are recomputed by the oracle (about 20 s of host time each)."""
    thth, to = env
    from scintools_amd.synth import arc_dynspec
    size = 4096
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=3, nimg=64)
    dyn -= dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    etas = np.geomspace(0.25, 4.0, 12) * eta_true
    CS = to.conjugate_spectrum(dyn, 0)
    N = float(dyn.size)
    chis, info = thth.chisq_sweep(dyn, thth.to_device(CS), tau, fd, etas, edges, N, return_info=True)
    assert np.all(info["status"] == 0) and len(set(int(v) for v in info["N"])) >= 3
    for i in (0, 4, 7, 11):
        ref = to.chisq_calc(dyn, CS, tau, fd, etas[i], edges, N)
        assert chis[i] == pytest.approx(ref, rel=1e-9), (i, etas[i] / eta_true)


def test_diagonal_back_map_vs_general_kernel_and_oracle(env, case, monkeypatch):
    """Round 6: the rank-1 Hermitian back-map on a uniform theta grid is rev_diag_kernel (whole diagonals of a Doppler column,
    plain float64 sums in one fixed order) instead of rev_gather_kernel (a partner search per theta_i, order-independent split
    sums).  At full size, for a flat, the true and a cropped steep curvature and a negative one: the call says which kernel ran;
    the two images hold the same pixels (identical empty-bin masks) and agree to 1e-12 of the peak (each pixel is the same
    addends in another order); a second call gives the same BITS; and at 2048^2 the image is the oracle's np.histogram2d image
    to 1e-12 as well (the oracle at 4096^2 is covered by test_modeler_vs_oracle at one curvature)."""
    import torch
    thth, to = env
    c = case
    size = c["size"]
    eta_true = c["eta"] / 0.93
    grid = thth._Grid(c["tau"], c["fd"], c["edges"])
    rng = np.random.default_rng(size)
    for factor in (0.2, 1.0, 3.5, -0.6):
        eta = factor * eta_true
        keep = grid.keep(abs(eta)) if factor > 0 else np.arange(grid.M, dtype=np.int32)
        th_red = thth._theta_centres(grid.edges_red(keep)) if factor > 0 else grid.th_cents
        n = len(th_red)
        v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.exp(-np.linspace(-2, 2, n) ** 2)
        w = np.array([3.3])
        th_t, v_t, w_t = thth.to_device(th_red, torch.float64), thth.to_device(v), thth.to_device(w, torch.float64)
        info = {}
        monkeypatch.setenv("SCINT_REV_DIAG", "1")
        a = thth._rev_map_dev(grid.geom, th_t, n, eta, True, vec_t=v_t, w_t=w_t, info=info).cpu().numpy()
        again = thth._rev_map_dev(grid.geom, th_t, n, eta, True, vec_t=v_t, w_t=w_t).cpu().numpy()
        assert info["uniform_grid"] == 1, factor
        assert np.array_equal(a, again), factor
        monkeypatch.setenv("SCINT_REV_DIAG", "0")
        b = thth._rev_map_dev(grid.geom, th_t, n, eta, True, vec_t=v_t, w_t=w_t, info=info).cpu().numpy()
        assert info["uniform_grid"] == 0, factor
        assert np.count_nonzero(a) > 1000, factor
        _assert_image_close(a, b, 1e-12, f"diagonal against general kernel, {factor} eta_true, N = {n}")
        if size == 2048 and factor in (0.2, 3.5):
            edges_red = grid.edges_red(keep)
            ref = np.nan_to_num(to.rev_map(np.outer(v, np.conj(v)) * w[0], c["tau"], c["fd"], eta, edges_red, True))
            _assert_image_close(a, ref, 1e-12, f"diagonal kernel against the oracle, {factor} eta_true")


def test_chisq_from_the_back_map_accumulators_fullsize(env, case, monkeypatch):
    """Round 6: on the symmetric axes of the path chi^2 of every uniform-grid curvature is formed by the back-map workgroups
    (interior pixels; column 0 and row 0 by the partner formula) and the image is not written.  At full size, fifteen curvatures
    from a narrow band to the whole axis and into the crop: the call reports the fused route, and chi^2 equals the
    written-image route's (SCINT_CHISQ_FUSE=0: round 4's Parseval pass) to 1e-12 -- which test_chisq_sweep_vs_chisq_calc holds
    to the oracle."""
    thth, to = env
    c = case
    dyn = c["dyn"]
    eta_true = c["eta"] / 0.93
    etas = np.concatenate([np.linspace(0.2, 1.4, 12), [2.2, 3.0, 3.8]]) * eta_true * 1.000123
    cs_t = thth.to_device(c["CS"])
    monkeypatch.setenv("SCINT_CHISQ_FUSE", "1")
    a, ia = thth.chisq_sweep(dyn, cs_t, c["tau"], c["fd"], etas, c["edges"], float(dyn.size), return_info=True)
    monkeypatch.setenv("SCINT_CHISQ_FUSE", "0")
    b, ib = thth.chisq_sweep(dyn, cs_t, c["tau"], c["fd"], etas, c["edges"], float(dyn.size), return_info=True)
    assert ia["fused"] and not ib["fused"] and ib["redone"] == 0
    assert ia["redone"] <= 3
    assert np.all(ia["status"] == 0) and np.all(np.isfinite(a))
    np.testing.assert_allclose(a, b, rtol=1e-12)
