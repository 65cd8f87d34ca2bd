"""The kernel SOURCES, interpreted on the host (tests/emu), against the oracle -- runs without a GPU.

What this is: scintools_amd/csrc/*.hip compiled for x86 on top of a small workgroup interpreter
(fibers for threads, 64-lane waves, barriers and cross-lane operations as scheduling points) and
driven through the same C ABI and the same Python wrappers as the GPU library.  It checks the
control flow and arithmetic of the kernels and of the host-side sweep scheduler in the build
container.  What it is not: a product path (scintools_amd never loads it and still raises without
a GPU) or parity evidence for the GPU (that is `pytest -m gpu`, which runs the gfx950 build of the
same sources; the two agree to rounding because they ARE the same sources).
Sizes are tiny: the interpreter runs ~10^4 times slower than the GPU.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture()
def emu(monkeypatch):
    import subprocess
    import emulated
    try:
        emulated.install(monkeypatch)
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:    # no usable clang++ on this machine
        pytest.skip(f"host interpreter could not be built: {exc}")
    from scintools_amd import ththmod
    return ththmod


@pytest.fixture(scope="module")
def to():
    from oracle import thth_oracle
    return thth_oracle


@pytest.fixture(scope="module")
def case():
    """A 96 x 80 arc dynamic spectrum with its axes, edges and a few curvatures."""
    from oracle import thth_oracle
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(96, 80, seed=5, nimg=12)
    dyn = dyn - dyn.mean()
    fd = thth_oracle.fft_axis(times, 1000.0, 1)
    tau = thth_oracle.fft_axis(freqs, 1.0, 1)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 100)
    etas = np.geomspace(0.5, 2.0, 5) * eta_true
    CS = thth_oracle.conjugate_spectrum(dyn, 1)
    return dict(dyn=dyn, fd=fd, tau=tau, edges=edges, etas=etas, CS=CS)


def test_interpreted_library_is_not_the_product(emu):
    from scintools_amd import _lib
    assert os.path.basename(_lib.load()._name) == "libscint_emu_test.so"
    assert "tests" in _lib.load()._name.split(os.sep)


def test_interpreter_semantics(emu):
    """The interpreter's own model of the hardware: cross-lane operations incl. a ballot that only
    part of the wave executes, a block barrier across four waves, and v_mfma_f64_16x16x4 with the
    operand layout of the programming guide (A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j,
    result register r of lane l = row (l >> 4) + 4 r, column l & 15)."""
    import ctypes
    import emulated
    lib = emulated.load()
    rng = np.random.default_rng(1)
    a, b, c = rng.standard_normal(64), rng.standard_normal(64), rng.standard_normal(256)
    d = np.zeros(256)
    ptr = lambda x: ctypes.c_void_p(x.ctypes.data)
    lib.emu_selftest_mfma(ptr(a), ptr(b), ptr(c), ptr(d))
    A = a.reshape(4, 16).T                      # A[i][k] = a[16 k + i]
    B = b.reshape(4, 16)                        # B[k][j] = b[16 k + j]
    C = np.zeros((16, 16))
    lanes, regs = np.meshgrid(np.arange(64), np.arange(4), indexing="ij")
    C[(lanes >> 4) + 4 * regs, lanes & 15] = c.reshape(64, 4)
    D = C + A @ B
    np.testing.assert_allclose(d.reshape(64, 4), D[(lanes >> 4) + 4 * regs, lanes & 15], rtol=1e-12, atol=1e-14)
    x = rng.standard_normal(256)
    out = np.zeros(320)
    lib.emu_selftest_wave(ptr(x), ptr(out), ctypes.c_int(40))
    l = np.arange(64)
    np.testing.assert_allclose(out[:64], x[:64].sum(), rtol=1e-13)
    assert np.array_equal(out[64:128], x[(l * 7) % 64])
    assert np.all(out[128:192] == float(int(x[5] * 1000.0)))
    assert np.array_equal(out[192:256], np.where(l < 40, np.sum(np.arange(40) % 3 == 0), -1))
    assert np.array_equal(out[256:320], x[64:128])


@pytest.mark.parametrize("hermetian", [True, False])
def test_gather_bit_equal(emu, to, case, hermetian):
    c = case
    for eta in c["etas"][[0, 2, 4]]:
        got, e_got = emu.thth_redmap(c["CS"], c["tau"], c["fd"], eta, c["edges"], hermetian)
        ref, e_ref = to.thth_redmap(c["CS"], c["tau"], c["fd"], eta, c["edges"], hermetian)
        assert np.array_equal(got, ref, equal_nan=True)
        assert np.array_equal(e_got, e_ref)


@pytest.mark.parametrize("nf,nt,npad", [(96, 80, 1), (64, 64, 0), (50, 20, 2)])
def test_conjugate_spectrum(emu, to, nf, nt, npad):
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, _, _ = arc_dynspec(nf, nt, seed=nf + nt, nimg=6)
    tau = to.fft_axis(freqs, 1.0, npad)
    mask = 2.5 * (tau[1] - tau[0])
    ref = to.conjugate_spectrum(dyn, npad, tau, mask)
    got = emu.conjugate_spectrum(dyn, npad, tau, mask, True).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_secondary_spectrum(emu):
    import torch
    from oracle import sspec_oracle
    from scintools_amd.dynspec import sspec_device
    from scintools_amd.synth import arc_dynspec
    dyn = arc_dynspec(64, 96, seed=2, nimg=6)[0]
    sec = sspec_device(emu.to_device(dyn, torch.float64)).cpu().numpy()
    ref = sspec_oracle.calc_sspec(dyn, 30.0, 1.0)[2]
    lin, lref = 10 ** (sec / 10), 10 ** (ref / 10)
    assert np.abs(lin - lref).max() <= 1e-10 * lref.max()


@pytest.mark.parametrize("nf,nt", [(130, 200), (257, 131), (200, 300)])
@pytest.mark.parametrize("kw", [{}, {"prewhite": True}, {"window": None}])
def test_secondary_spectrum_two_trip_path(emu, nf, nt, kw):
    """Shapes the strided-axis-first path of sspec.hip takes (next_pow2 of both axes >= 256): odd and even
    nt (16-byte and 8-byte input loads), nf just above a power of two, all three source variants."""
    import torch
    from oracle import sspec_oracle
    from scintools_amd.dynspec import sspec_device
    rng = np.random.default_rng(nf * 1000 + nt)
    dyn = rng.standard_normal((nf, nt)) + 3.0 + 0.5 * np.sin(0.07 * np.arange(nt))[None, :] * np.cos(0.11 * np.arange(nf))[:, None]
    sec = sspec_device(emu.to_device(dyn, torch.float64), **kw).cpu().numpy()
    ref = sspec_oracle.calc_sspec(dyn, 30.0, 1.0, **kw)[2]
    assert sec.shape == ref.shape
    lin, lref = 10 ** (sec / 10), 10 ** (ref / 10)
    assert np.abs(lin - lref).max() <= 1e-10 * lref.max()
    strong = lref > 1e-6 * lref.max()
    assert np.abs(sec - ref)[strong].max() <= 1e-8


@pytest.mark.parametrize("grid", [1, 3, 8])
@pytest.mark.parametrize("kw", [{}, {"prewhite": True}])
def test_secondary_spectrum_persistent_kernels_do_not_depend_on_the_grid(emu, monkeypatch, grid, kw):
    """calc_sspec's persistent kernels (sspec.hip: a workgroup walks over rows / column pairs, the next one prefetched into the
    registers the current one leaves, a row's stores issued one iteration later): with 1, 3 and 8 workgroups (several iterations
    each, with and without the XCD remap of whole rounds) the result equals the default launch's bit for bit, short rows and
    columns (zeroed tails) included."""
    import torch
    from scintools_amd.dynspec import sspec_device
    rng = np.random.default_rng(7)
    dyn = rng.standard_normal((300, 520)) + 2.0
    monkeypatch.delenv("SCINT_SSPEC_MAXGRID", raising=False)
    ref = sspec_device(emu.to_device(dyn, torch.float64), **kw).cpu().numpy()
    monkeypatch.setenv("SCINT_SSPEC_MAXGRID", str(grid))
    got = sspec_device(emu.to_device(dyn, torch.float64), **kw).cpu().numpy()
    assert np.array_equal(got, ref)


def test_secondary_spectrum_zero_subnormal_and_nonfinite_powers(emu):
    """The branch-free logarithm of the row kernel hands zero, subnormal and non-finite powers to the series form: -inf, NaN and
    the subnormal values come out where NumPy puts them."""
    import warnings
    import torch
    from oracle import sspec_oracle
    from scintools_amd.dynspec import sspec_device
    rng = np.random.default_rng(1)
    nan_in = 1.0 + rng.standard_normal((130, 200))
    nan_in[3, 177] = np.nan
    for dyn in (np.full((130, 200), 2.5), nan_in, 1e-160 * rng.standard_normal((130, 200))):
        sec = sspec_device(emu.to_device(dyn, torch.float64)).cpu().numpy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = sspec_oracle.calc_sspec(dyn, 30.0, 1.0)[2]
        assert np.array_equal(np.isneginf(sec), np.isneginf(ref)) and np.array_equal(np.isnan(sec), np.isnan(ref))
        fin = np.isfinite(ref)
        if fin.any():
            assert np.abs(sec - ref)[fin].max() <= 1e-8


def test_device_crop_tables_equal_the_host_ones(emu, case):
    """scint_sweep_keep against the NumPy expression of thth_redmap's crop (ththmod.py:153-155): same indices,
    same counts, for curvatures from 'keeps everything' to 'keeps nothing'."""
    grid = emu._Grid(case["tau"], case["fd"], np.linspace(-case["fd"].max(), case["fd"].max(), 700))
    etas = np.concatenate((np.geomspace(1e-3, 1e3, 300) * case["etas"][2], [1e12 * case["etas"][2]]))
    keep_t, n_dev = emu._sweep_inputs_dev(grid, etas)
    keep_h, n_h = emu._sweep_inputs(grid, etas)
    assert np.array_equal(n_dev, n_h) and n_h.max() > 300 and n_h.min() <= 1     # more than one 256-element scan step; the |theta| < fd_max/2 half of the crop binds too
    kd = keep_t.cpu().numpy()
    for i, n in enumerate(n_h):
        assert np.array_equal(kd[i, :n], keep_h[i, :n])


@pytest.mark.parametrize("size", [192, 300])
def test_eigenvalue_sweep_several_block_rows(emu, to, size):
    """Matrices of 3 and 5 block rows: pairs of rows per mat-vec workgroup with strips beyond the first column
    range, a second row that starts one tile late, an unpaired last row (the `case` fixture stays below 128)."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=9, nimg=6, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    CS = to.conjugate_spectrum(dyn, 0)
    etas = np.array([0.8, 1.0, 1.3]) * eta_true
    ref = np.array([to.Eval_calc(CS, tau, fd, e, edges) for e in etas])
    eigs, info = emu.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
    assert np.all(info["status"] == 0) and info["N"].max() == size - 1
    np.testing.assert_allclose(eigs, ref, rtol=1e-9)
    w, V, _ = emu.eigvec_sweep(CS, tau, fd, etas[:1], edges)
    red, _ = to.thth_redmap(CS, tau, fd, etas[0], edges)
    v = V[0, : red.shape[0]].cpu().numpy()
    assert np.linalg.norm(red @ v - w[0] * v) <= 1e-8 * abs(w[0])


def test_eigenvalue_sweep_vs_arpack(emu, to, case):
    """The batched two-vector Lanczos sweep incl. its two-stream scheduler (run here in enqueue
    order) against the oracle's ARPACK eigsh, and independent of the batch size."""
    c = case
    eigs, info = emu.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"], c["edges"], return_info=True)
    ref = np.array([to.Eval_calc(c["CS"], c["tau"], c["fd"], e, c["edges"]) for e in c["etas"]])
    assert np.all(info["status"] == 0)
    np.testing.assert_allclose(eigs, ref, rtol=1e-9)
    eigs2 = emu.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"], c["edges"], batch=2)
    assert np.array_equal(eigs, eigs2)


@pytest.fixture()
def mixed(emu):
    """The eigenvalue sweeps of this test iterate on the complex64 copy and certify on the complex128 tiles."""
    emu.sweep_precision("mixed")
    yield emu
    emu.sweep_precision("f64")


def _sweep_stats(thth):
    import ctypes
    from scintools_amd import _lib
    st = (ctypes.c_double * 4)()
    _lib.check(_lib.load().scint_sweep_stats(st), "scint_sweep_stats")
    return dict(bytes32=st[0], bytes64=st[1], certified=st[2], cert_passes=st[3])


@pytest.mark.parametrize("size", [96, 192, 300, 520])
def test_mixed_sweep_against_float64_sweep_and_arpack(emu, to, size):
    """Mixed precision (eigen_packed.hip): the passes stream complex64 tiles -- four block rows per workgroup, rows that
    start one to three tiles late, a last group of one to three rows, strips of two and four tiles --, the returned
    value is the Ritz value of the certificate pass on the complex128 tiles.  It has to agree with the float64 sweep far
    inside its tolerance (both are Ritz values of the SAME float64 matrix under the same bound) and with ARPACK to the
    parity bar; every curvature goes through exactly one certificate, which here passes at its first step."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=9, nimg=6, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    CS = to.conjugate_spectrum(dyn, 0)
    etas = np.array([0.8, 1.0, 1.3]) * eta_true if size < 500 else np.array([1.0]) * eta_true
    e64, i64 = emu.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
    assert _sweep_stats(emu)["bytes32"] == 0
    assert emu.sweep_precision("mixed") == "f64"
    try:
        emx, imx = emu.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
        st = _sweep_stats(emu)
        emx2 = emu.eval_sweep(CS, tau, fd, etas, edges, batch=2)
    finally:
        assert emu.sweep_precision("f64") == "mixed"
    assert np.all(imx["status"] == 0) and np.all(i64["status"] == 0)
    np.testing.assert_allclose(emx, e64, rtol=1e-13)
    if size < 500:
        ref = np.array([to.Eval_calc(CS, tau, fd, e, edges) for e in etas])
        np.testing.assert_allclose(emx, ref, rtol=1e-9)
    assert np.array_equal(emx, emx2)                                  # batch size / slot grouping: same bits
    assert st["certified"] == len(etas) and st["cert_passes"] == len(etas)
    n_ = imx["N"].astype(float)
    assert st["bytes64"] == np.sum(8 * n_ * (n_ + 1))                 # ONE complex128 pass per curvature
    assert st["bytes32"] == np.sum(4 * n_ * (n_ + 1) * (imx["iters"] - 1))
    assert np.all(imx["iters"] <= i64["iters"] + 3)                   # the iteration phase costs what the float64 sweep costs (tol / 4: a check or so more)


def test_mixed_sweep_does_not_depend_on_the_units_of_the_data(mixed, to, case):
    """The complex64 copy is taken from the spectrum times a power of two derived from max |CS|: data scaled by 2^-200
    (every element far below the float32 range) or 2^+150 (far above it) give the same bits, scaled."""
    c = case
    base = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"], c["edges"])
    for k in (-200, 150):
        got, info = mixed.eval_sweep(c["CS"] * 2.0 ** k, c["tau"], c["fd"], c["etas"], c["edges"], return_info=True)
        assert np.all(info["status"] == 0)
        assert np.array_equal(got, np.ldexp(base, k))


def test_mixed_sweep_edge_cases(mixed, to, case):
    """What the float64 sweep does at the edges of the domain, the mixed one does too: an all-zero spectrum gives 0,
    a crop to nothing NaN (status EMPTY), non-finite input NaN, an iteration cap a NOCONV status (NaN, like the
    reference's exception path), tiny matrices -- whose Krylov space is complete before anything converges -- LAPACK's
    value."""
    c = case
    assert mixed.eval_sweep(np.zeros_like(c["CS"]), c["tau"], c["fd"], c["etas"][:1], c["edges"])[0] == 0.0
    etas = np.array([c["etas"][2], 1e9 * c["etas"][2]])
    eigs, info = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], etas, c["edges"], return_info=True)
    assert np.isfinite(eigs[0]) and np.isnan(eigs[1]) and info["status"][1] == 5
    bad = c["CS"].copy()
    bad[bad.shape[0] // 2 + 3, bad.shape[1] // 2 + 5] = np.nan
    ref64 = None
    mixed.sweep_precision("f64")
    ref64, i64 = mixed.eval_sweep(bad, c["tau"], c["fd"], c["etas"][2:3], c["edges"], return_info=True)
    mixed.sweep_precision("mixed")
    got, info = mixed.eval_sweep(bad, c["tau"], c["fd"], c["etas"][2:3], c["edges"], return_info=True)
    assert info["status"][0] == i64["status"][0] and np.array_equal(np.isnan(got), np.isnan(ref64))
    eigs, info = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"][2:3], c["edges"], max_iter=3, return_info=True)
    assert info["status"][0] == 4 and np.isnan(eigs[0])
    for nedge in (4, 6, 10, 34):
        edges = np.linspace(-c["fd"].max() / 2, c["fd"].max() / 2, nedge)
        red, _ = to.thth_redmap(c["CS"], c["tau"], c["fd"], c["etas"][2], edges)
        eig, info = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"][2:3], edges, return_info=True)
        assert info["N"][0] == red.shape[0] and info["status"][0] == 0
        assert eig[0] == pytest.approx(abs(np.linalg.eigvalsh(red)[-1]), rel=1e-10, abs=1e-9)


def test_mixed_sweep_certificate_measures_the_float64_residual(mixed, to, case):
    """The certificate must SEE the residual the complex64 copy leaves (|| A v - theta v || ~ 1e-8 |theta|), although it is
    1e-8 of the vectors it is computed from: formed as W^H W - A^H A it would vanish in rounding (and under the pivot
    floor) and every certificate would pass.  With a tolerance below what that residual allows (tol = 1e-17 asks for
    resid < 3e-10 |theta| at these gaps) the certificates must therefore NOT pass at their first step: the runs continue on
    the complex128 tiles, and still end with the float64 sweep's values."""
    c = case
    mixed.sweep_precision("f64")
    ref, iref = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"], c["edges"], tol=1e-17, return_info=True)
    mixed.sweep_precision("mixed")
    got, info = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"], c["edges"], tol=1e-17, return_info=True)
    st = _sweep_stats(mixed)
    assert np.all(info["status"] == 0) and np.all(iref["status"] == 0)
    np.testing.assert_allclose(got, ref, rtol=1e-13)
    assert st["certified"] == len(c["etas"]) and st["cert_passes"] >= 2 * len(c["etas"])
    # at the default tolerance the same certificates pass at once (the residual is small enough, and is seen to be)
    got, info = mixed.eval_sweep(c["CS"], c["tau"], c["fd"], c["etas"], c["edges"], return_info=True)
    st = _sweep_stats(mixed)
    assert st["cert_passes"] == len(c["etas"])


def test_mixed_sweep_fuzz_against_the_float64_sweep(emu, to):
    """Random small problems (shapes, padding, edge counts either side of the 64-row blocks, 1 to 19 images, noise from
    none to five times the signal, curvatures a factor five either side of the arc, batch sizes): the mixed sweep returns
    the float64 sweep's status for every curvature and its value to 1e-11 (100 such trials ran clean when this was written;
    the test keeps twelve)."""
    from scintools_amd.synth import arc_dynspec
    rng = np.random.default_rng(11)
    for trial in range(12):
        nf, nt = int(rng.integers(40, 260)), int(rng.integers(40, 260))
        nedge = int(rng.choice([34, 64, 66, 100, 128, 130, 192, 194, 200, 256, 258, 260]))
        npad = int(rng.integers(0, 2))
        dyn, freqs, times, eta_true = arc_dynspec(nf, nt, seed=int(rng.integers(1 << 30)), nimg=int(rng.integers(1, 20)),
                                                  noise=float(rng.choice([0.0, 0.05, 1.0, 5.0])))
        dyn = dyn - dyn.mean()
        fd, tau = to.fft_axis(times, 1000.0, npad), to.fft_axis(freqs, 1.0, npad)
        edges = np.linspace(-fd.max() / 2, fd.max() / 2, nedge)
        CS = to.conjugate_spectrum(dyn, npad)
        etas = eta_true * np.exp(rng.uniform(np.log(0.2), np.log(5.0), size=int(rng.integers(1, 7))))
        batch = int(rng.integers(1, 8))
        e64, i64 = emu.eval_sweep(CS, tau, fd, etas, edges, return_info=True, batch=batch)
        emu.sweep_precision("mixed")
        try:
            emx, imx = emu.eval_sweep(CS, tau, fd, etas, edges, return_info=True, batch=batch)
        finally:
            emu.sweep_precision("f64")
        assert np.array_equal(i64["status"], imx["status"]), (trial, i64["status"], imx["status"])
        good = i64["status"] == 0
        np.testing.assert_allclose(emx[good], e64[good], rtol=1e-11, err_msg=f"trial {trial}")


@pytest.mark.parametrize("strip_len", [5, 14])
def test_mixed_sweep_long_strips(strip_len):
    """Strips of 5 and of 14 column tiles (what 4096^2 runs with) x four block rows on a 700^2 problem (11 block rows:
    groups of 4, 4 and 3 rows; strips of full length, remainders of one and two tiles; the two-tiles-at-a-time loop of the
    complex64 mat-vec with even and odd counts and rows that start one to three tiles late): the mixed sweep against the
    float64 sweep with the same forced strip length (SCINT_STRIP_LEN is read once per process: a process of its own)."""
    import json
    import subprocess
    try:
        import emulated
        emulated.load()
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:
        pytest.skip(f"host interpreter could not be built: {exc}")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "strip_probe.py")
    env = dict(os.environ, SCINT_STRIP_LEN=str(strip_len), OPENBLAS_NUM_THREADS="1")
    out = subprocess.run([sys.executable, probe, "700"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["f64"]["status"] == [0, 0] and d["mixed"]["status"] == [0, 0] and max(d["f64"]["N"]) == 699
    np.testing.assert_allclose(d["mixed"]["eigs"], d["f64"]["eigs"], rtol=1e-12)
    assert d["mixed"]["stats"][2] == 2 and d["mixed"]["stats"][3] == 2          # two certificates, one pass each
    assert d["mixed"]["stats"][0] > 0 and d["f64"]["stats"][0] == 0


def test_mixed_sweep_small_gaps_and_several_spectra(mixed, to):
    """Noise-like spectra (small spectral gaps; the certificate's gap comes from the SECOND Ritz vector of the iteration
    phase) against LAPACK, and the many-spectra entry point (scint_eval_sweep_multi: one power-of-two scale per
    spectrum -- the second spectrum is 2^40 times the first) against the single-spectrum one."""
    rng = np.random.default_rng(5)
    nf, nt = 150, 130
    dyn = rng.standard_normal((nf, nt)) + 5 * np.outer(np.cos(np.arange(nf) * 0.3), np.cos(np.arange(nt) * 0.2))
    fd = to.fft_axis(np.arange(nt) * 30.0, 1000.0, 0)
    tau = to.fft_axis(1400 + np.arange(nf) * 0.1, 1.0, 0)
    CS = to.conjugate_spectrum(dyn - dyn.mean(), 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 120)
    etas = tau.max() / (fd.max() / 2) ** 2 * np.array([0.3, 1.0, 3.0])
    ref = np.array([np.linalg.eigvalsh(to.thth_redmap(CS, tau, fd, e, edges)[0])[-1] for e in etas])
    eigs, info = mixed.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
    assert np.all(info["status"] == 0)
    np.testing.assert_allclose(eigs, np.abs(ref), rtol=1e-11)
    import torch
    stack = torch.from_numpy(np.stack([CS, CS * 2.0 ** 40]))
    out = mixed.eval_sweep_multi(stack, [(tau, fd, edges)] * 2, [etas, etas[::-1]])
    assert np.array_equal(out[0], eigs) and np.array_equal(out[1], np.ldexp(eigs[::-1], 40))


@pytest.mark.parametrize("size", [96, 192, 300])
def test_mixed_all_eigenpair_sweeps_against_the_float64_ones(emu, to, size):
    """``sweep_precision("mixed-all")`` (scint_sweep_precision(2)): the eigenPAIR sweeps iterate on the complex64 copy to
    the eigenvalue rule and finish the vector on the complex128 tiles from the two Ritz vectors, to the float64 sweep's
    own residual rule.  Eigenvectors (up to the one global phase), eigenvalues
    and chi^2 equal the float64 sweep's far inside the parity bar; fewer complex128 bytes are streamed."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(size, size, seed=9, nimg=6, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, size)
    CS = to.conjugate_spectrum(dyn, 0)
    etas = np.array([0.8, 1.0, 1.3]) * eta_true
    w64, v64 = emu.eigvec_sweep(CS, tau, fd, etas, edges)[:2]
    b64 = _sweep_stats(emu)
    c64 = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 1.0)
    assert b64["bytes32"] == 0
    assert emu.sweep_precision("mixed-all") == "f64"
    try:
        wmx, vmx = emu.eigvec_sweep(CS, tau, fd, etas, edges)[:2]
        bmx = _sweep_stats(emu)
        cmx = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 1.0)
        emx = emu.eval_sweep(CS, tau, fd, etas, edges)               # mode 2 includes the mixed eigenvalue sweep
    finally:
        assert emu.sweep_precision("f64") == "mixed-all"
    np.testing.assert_allclose(wmx, w64, rtol=1e-12)
    np.testing.assert_allclose(emx, w64, rtol=1e-12)
    for a, b in zip(vmx.cpu().numpy(), v64.cpu().numpy()):
        ph = np.vdot(b, a) / abs(np.vdot(b, a))                      # rows are unit vectors of arbitrary phase
        assert np.abs(a / ph - b).max() <= 1e-10 * np.abs(b).max()   # (1 - |<a, b>| would only see the SQUARE of this)
    np.testing.assert_allclose(cmx, c64, rtol=1e-9)
    assert bmx["bytes32"] > 0 and bmx["certified"] == len(etas)
    assert bmx["bytes64"] + bmx["bytes32"] < b64["bytes64"]          # what it is for


def test_modeler_and_chisq_sweep(emu, to, case):
    c = case
    eta = c["etas"][2]
    got = emu.modeler(c["CS"], c["tau"], c["fd"], eta, c["edges"])
    ref = to.modeler(c["CS"], c["tau"], c["fd"], eta, c["edges"])
    assert np.array_equal(got[0], ref[0])                                  # thth_red
    assert abs(got[5] - ref[5]) <= 1e-9 * abs(ref[5])                      # w
    for k in (2, 3):                                                        # recov, model
        g, r = np.nan_to_num(got[k]), np.nan_to_num(ref[k])
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k]))
        assert np.abs(g - r).max() <= 1e-9 * np.abs(r).max()
    chis = emu.chisq_sweep(c["dyn"], c["CS"], c["tau"], c["fd"], c["etas"][1:4], c["edges"], 1.0)
    refc = np.array([to.chisq_calc(c["dyn"], c["CS"], c["tau"], c["fd"], e, c["edges"], 1.0) for e in c["etas"][1:4]])
    np.testing.assert_allclose(chis, refc, rtol=1e-9)


@pytest.mark.parametrize("nf,nt", [(96, 80), (97, 81), (64, 90)])
def test_chisq_sweep_by_parseval_against_the_oracle(emu, to, nf, nt):
    """npad = 0, no mask, finite dspec: scint_chisq_sweep takes chi^2 from recov and fft2(dspec) by Parseval's identity
    (chisq_parseval_kernel) instead of transforming the model back -- even and odd axis lengths (the partner index
    2 h - p mod P), against the oracle's chisq_calc (the reference's time-domain sum, ththmod.py:330-368) and against
    the product's own per-eta chisq_calc, which still goes through the model.  A NaN pixel in dspec sends the sweep down
    the model route (chisq_calc's default mask is then not all-true): same bar."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(nf, nt, seed=11, nimg=8, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 70)
    CS = to.conjugate_spectrum(dyn, 0)
    assert CS.shape == dyn.shape
    etas = np.array([0.7, 1.0, 1.6]) * eta_true
    got = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0)
    ref = np.array([to.chisq_calc(dyn, CS, tau, fd, e, edges, 3.0) for e in etas])
    np.testing.assert_allclose(got, ref, rtol=1e-9)
    one = np.array([emu.chisq_calc(dyn, CS, tau, fd, e, edges, 3.0) for e in etas])
    np.testing.assert_allclose(got, one, rtol=1e-11)
    assert np.array_equal(got, emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0))
    holed = dyn.copy()
    holed[3, 5] = np.nan
    got_h = emu.chisq_sweep(holed, CS, tau, fd, etas, edges, 3.0)
    ref_h = np.array([to.chisq_calc(holed, CS, tau, fd, e, edges, 3.0) for e in etas])
    np.testing.assert_allclose(got_h, ref_h, rtol=1e-9)


def test_chisq_sweep_tail_batches_and_delay_band(emu, to):
    """Round 5: the chi^2 sweep hands the curvatures a chunk retires to its tail in batches (four launches per <= 8 curvatures
    instead of eleven API calls per curvature), and back-map and chi^2 touch only the delay rows a curvature can reach
    (|tau| <= |eta| max theta^2, made symmetric about tau = 0); the rows outside contribute sum |fft2(dspec)|^2 from prefix sums.
    Two delay slabs (ntau = 1100), curvatures from a band of a few rows to the whole axis and into the crop, more curvatures
    than one batch holds: against the oracle's chisq_calc (1e-9), against the product's per-eta chisq_calc through the model
    transform (1e-11), and bit-identical whatever the batch a curvature lands in (subset, reversed order, one slot group)."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(1100, 48, seed=21, nimg=8, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 60)
    CS = to.conjugate_spectrum(dyn, 0)
    th = emu._Grid(tau, fd, edges).th_cents
    eta_full = np.abs(tau).max() / (th**2).max()                    # tau_map reaches the end of the delay axis here
    etas = np.array([0.004, 0.03, 0.11, 0.3, 0.45, 0.49, 0.51, 0.7, 0.95, 1.05, 1.6, 3.0, 0.2, 0.6, 0.02, 0.8, 2.0, 0.25]) * eta_full
    got = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0)
    assert np.all(np.isfinite(got))
    sample = [0, 2, 5, 6, 9, 11]
    ref = np.array([to.chisq_calc(dyn, CS, tau, fd, etas[i], edges, 3.0) for i in sample])
    np.testing.assert_allclose(got[sample], ref, rtol=1e-9)
    one = np.array([emu.chisq_calc(dyn, CS, tau, fd, e, edges, 3.0) for e in etas])
    np.testing.assert_allclose(got, one, rtol=1e-11)
    assert np.array_equal(got[::-1], emu.chisq_sweep(dyn, CS, tau, fd, etas[::-1], edges, 3.0))
    assert np.array_equal(got[3:9], emu.chisq_sweep(dyn, CS, tau, fd, etas[3:9], edges, 3.0))
    assert np.array_equal(got, emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, batch=3))
    # the claim behind the band, on the full image of the one-image back-map: no pair falls outside the rows
    # bin(-Y) .. bin(Y), Y = eta max theta^2 (so nothing is lost by not computing them)
    import torch
    for i in (0, 1, 3, 8):
        red, V, w, recov, model, keep = emu._modeler_dev(emu.to_device(CS, torch.complex128), emu._Grid(tau, fd, edges), float(etas[i]))
        img = recov.numpy()
        thr = emu._theta_centres(emu._Grid(tau, fd, edges).edges_red(keep))
        Y = abs(etas[i]) * (thr**2).max()
        step = tau[1] - tau[0]
        rows = np.nonzero(np.abs(img).sum(axis=1))[0]
        lo, hi = np.floor((-Y - tau[0]) / step + 0.5), np.floor((Y - tau[0]) / step + 0.5)
        assert rows.min() >= lo - 1 and rows.max() <= hi + 1, (i, rows.min(), rows.max(), lo, hi)
        if i < 3:
            assert hi - lo < 0.5 * len(tau)          # (these curvatures do have a narrow band: the test is not vacuous)


@pytest.mark.parametrize("irregular", [False, True])
def test_chisq_sweep_shares_the_backmap_walk_between_same_crop_curvatures(emu, to, irregular):
    """Round 5: which theta_j pair with theta_i in a Doppler column does not depend on the curvature, so the curvatures of a
    sweep that keep the same theta centres share ONE partner table (launch_rev_walk_table) instead of each walking the
    centres in its back-map.  Same pairs, order-independent sums: chi^2 must be BIT-identical with the table (groups of >= 8
    same-crop curvatures) and without it (share_walk=False), on a uniform theta grid and on an irregular one, where some
    columns' windows do not bracket the column and keep the in-kernel walk; and equal to the oracle's chisq_calc."""
    from scintools_amd.synth import arc_dynspec
    # (theta spacing about half the Doppler step, as on the grids of the path: a window of W = 6 centres brackets every column of the
    #  uniform grid, and of 65 of the 96 columns of the irregular one -- the other 31 keep the in-kernel walk)
    dyn, freqs, times, eta_true = arc_dynspec(300, 96, seed=23, nimg=8, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 180)
    if irregular:
        rng = np.random.default_rng(3)
        edges = np.sort(edges + rng.uniform(-0.45, 0.45, edges.shape[0]) * (edges[1] - edges[0]))
    CS = to.conjugate_spectrum(dyn, 0)
    th = emu._Grid(tau, fd, edges).th_cents
    eta_full = np.abs(tau).max() / (th**2).max()
    # ten curvatures that keep every centre (one crop: a table), then four that crop differently (on their own)
    etas = np.concatenate([np.linspace(0.05, 0.95, 10), [1.3, 1.9, 2.6, 3.4]]) * eta_full
    keep_idx, keep_n = emu._sweep_inputs(emu._Grid(tau, fd, edges), etas)
    _, group = emu._reduced_centres(emu._Grid(tau, fd, edges), keep_idx, keep_n, return_groups=True)
    assert np.sum(group == group[0]) >= 8 and len(set(group.tolist())) >= 4
    shared = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0)
    alone = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, share_walk=False)
    assert np.all(np.isfinite(shared)) and np.array_equal(shared, alone)
    ref = np.array([to.chisq_calc(dyn, CS, tau, fd, etas[i], edges, 3.0) for i in (0, 5, 9, 12)])
    np.testing.assert_allclose(shared[[0, 5, 9, 12]], ref, rtol=1e-9)


def test_chisq_sweep_does_not_share_a_walk_table_on_a_false_promise(emu, to, monkeypatch):
    """scint_chisq_sweep takes the caller's crop_group on trust only as far as it can check -- members of a group must
    hold the same reduced centres, and the library compares their th_red rows on the device before it shares a partner table.  Here
    the wrapper is made to lie: one member of the ten-curvature group gets other centres (same count).  With the check that member
    (and its group) walk in the kernel, and chi^2 equals the run that shares nothing, bit for bit."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(300, 96, seed=23, nimg=8, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 180)
    CS = to.conjugate_spectrum(dyn, 0)
    th = emu._Grid(tau, fd, edges).th_cents
    eta_full = np.abs(tau).max() / (th**2).max()
    etas = np.concatenate([np.linspace(0.05, 0.95, 10), [1.3, 1.9, 2.6, 3.4]]) * eta_full
    honest = emu._reduced_centres_of_ranges

    def lying(grid, first, n):
        out = honest(grid, first, n)
        if out is None:
            return None
        th_red, group = out
        th_red = np.array(th_red)
        th_red[4, :n[4]] *= 0.75                     # member 4 of the big group: other centres, the same count, the same group
        return th_red, group
    monkeypatch.setattr(emu, "_reduced_centres_of_ranges", lying)
    shared = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0)
    alone = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, share_walk=False)
    assert np.all(np.isfinite(shared)) and np.array_equal(shared, alone)


@pytest.mark.parametrize("shape", [(3, 4, 16, 12), (2, 2, 64, 150), (3, 3, 128, 128), (2, 2, 126, 130), (1, 1, 8, 8)])
def test_device_mosaic_is_the_host_loop_bit_for_bit(emu, shape):
    """ththmod.mosaic_device (csrc/mosaic.hip) against ththmod.mosaic (the reference's loop, ththmod.py:1492-1554; itself pinned
    to the oracle's bit for bit): NumPy's summation order (8192-element buffer pieces, pairwise sums of <= 128-double runs) and
    its product arithmetic (fused multiply-adds of its SIMD loops; operands swapped by temporary elision from 256 KiB on --
    128 x 128 here) restated on the device, the scalar steps (mean, angle, exp) in NumPy itself: not a bit differs."""
    import torch
    rng = np.random.default_rng(sum(shape))
    ch = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 10.0 ** rng.integers(-2, 3, shape[:2] + (1, 1))
    assert emu._numpy_mosaic_modes(shape[2], shape[3]) is not None
    got = emu.mosaic_device(torch.from_numpy(ch)).numpy()
    assert np.array_equal(got, emu.mosaic(ch))


@pytest.mark.parametrize("fortran", [False, True])
@pytest.mark.parametrize("cw", [(64, 150), (128, 128), (16, 12), (20, 30)])
def test_device_chunk_cut_is_numpy_bit_for_bit(emu, cw, fortran):
    """ththmod.chunk_cut_device against the reference's three lines per chunk (dynspec.py:1782-1790: copy, subtract nanmean,
    nan_to_num) and the padding value (the chunk's mean, ththmod.py:783), NaNs in some windows: the same bits -- also for a
    Fortran-ordered dynamic spectrum (a transposed view, as files load), whose windows NumPy copies and sums column by column."""
    import torch
    rng = np.random.default_rng(5)
    dyn = rng.standard_normal((200, 300)) * 5 + 3
    dyn[5, 7] = np.nan
    dyn[100:120, 40] = np.nan
    if fortran:
        dyn = np.asfortranarray(dyn)
    cwf, cwt = cw
    org = [(0, 0), (10, 20), (200 - cwf, 300 - cwt), (min(90, 200 - cwf), 30)]
    out, pad = emu.chunk_cut_device(torch.from_numpy(np.ascontiguousarray(dyn)), org, cwf, cwt, fortran_order=fortran)
    for k, (r, c) in enumerate(org):
        d2 = np.copy(dyn[r:r + cwf, c:c + cwt])
        d2 -= np.nanmean(d2)
        d2 = np.nan_to_num(d2)
        assert np.array_equal(out[k].numpy(), d2) and float(pad[k]) == float(d2.mean())


def test_chunk_retrieval_in_byte_bounded_groups(emu, to, capsys):
    """the batched phase retrieval stacks conjugate spectra only up to a byte budget (groups, as the fit path
    does), and a chunk that cannot be prepared is left zero with its error printed while the others go on -- the reference's
    single_chunk_retrieval does the same chunk by chunk (ththmod.py:1471-1475).  Groups of one, of two and one group of
    all give the same chunks; each equals the oracle's single_chunk_retrieval up to its global phase."""
    from scintools_amd.synth import arc_dynspec
    chunks = []
    for k in range(3):
        dyn, freqs, times, eta_true = arc_dynspec(48, 40, seed=30 + k, nimg=8)
        dyn = dyn - dyn.mean()
        fd = to.fft_axis(times, 1000.0, 1)
        chunks.append((dyn, np.linspace(-fd.max() / 2, fd.max() / 2, 40), times, freqs, eta_true))
    bad = (chunks[0][0], chunks[0][1], chunks[0][2][:7], chunks[0][3], chunks[0][4])      # time axis of the wrong length
    everything = emu.chunk_retrieval_batch(chunks, 1, 0.0)
    one_by_one = emu.chunk_retrieval_batch(chunks, 1, 0.0, group_bytes=1)
    assert np.array_equal(everything, one_by_one)
    with_bad = emu.chunk_retrieval_batch([chunks[0], bad, chunks[1], chunks[2]], 1, 0.0, group_bytes=2 * 16 * 96 * 80)
    assert "Chunk 1:" in capsys.readouterr().out
    assert np.array_equal(with_bad[[0, 2, 3]], everything) and not with_bad[1].any()
    for k, (dyn, edges, times, freqs, eta) in enumerate(chunks):
        ref = to.single_chunk_retrieval(dyn, edges, times, freqs, eta, 1)
        got = everything[k] * np.exp(-1j * np.angle(np.vdot(ref, everything[k])))
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()


def test_rev_map_explicit_matrix(emu, to, case):
    c = case
    rng = np.random.default_rng(0)
    n = c["edges"].shape[0] - 1
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    herm = a + a.conj().T
    for hermetian, mat in ((True, herm), (False, a)):
        got = emu.rev_map(mat, c["tau"], c["fd"], c["etas"][2], c["edges"], hermetian)
        ref = to.rev_map(mat, c["tau"], c["fd"], c["etas"][2], c["edges"], hermetian)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.abs(np.nan_to_num(got) - np.nan_to_num(ref)).max() <= 1e-9 * np.abs(np.nan_to_num(ref)).max()


def test_back_map_bits_are_pinned(emu, monkeypatch):
    """rev_map images of the interpreted kernel on 29 grids (rank-1 and explicit, Hermitian or not, irregular theta,
    several delay slabs and 256-lane chunks, pairs pushed off the delay axis; odd axis lengths and axes that are not
    symmetric about 0 -- the seven digests added in round 4) have the SHA-256 they had before the round-3 rewrite of
    rev_gather_kernel (one copy of the pair arithmetic, chunk pre-pass, loop-free bin) and before round 4's changes (per-image
    constants from rev_setup_kernel, chunk-long pre-pass blocks, reciprocal table; the paired-column kernel that was
    measured and dropped produced the same bits too): the sums are order-independent by construction, so a rewrite may
    not change a bit."""
    import json
    import revmap_probe
    monkeypatch.setenv("SCINT_REV_DIAG", "0")        # round 6: rank-1 Hermitian images on a uniform grid have their own kernel (next test)
    got = revmap_probe.digests(revmap_probe.images(emu))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "revmap_bits.json")) as fh:
        want = json.load(fh)
    assert got == want, [k for k in want if got.get(k) != want[k]]


def test_diagonal_back_map_vs_oracle_and_pinned(emu, to):
    """The uniform-grid kernel of round 6 (rev_diag_kernel: pairs along diagonals, plain float64 sums in a fixed order) on the
    grids of the probe: every rank-1 image on a uniform grid is formed by it (the call's scratch says which kernel ran), equals
    the oracle's np.histogram2d image to 1e-12 of the peak with the same set of non-zero pixels (the same pairs in the same
    pixels; the order of a pixel's addends differs from NumPy's), and has the SHA-256 it had when the kernel was written
    (tests/golden/revmap_diag_bits.json: the sums have a fixed order, so a schedule may not change a bit); the irregular grids
    keep the general kernel and its pinned bits."""
    import json
    import hashlib
    import torch
    import revmap_probe
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "revmap_bits.json")) as fh:
        general = json.load(fh)
    with open(os.path.join(here, "golden", "revmap_diag_bits.json")) as fh:
        pinned = json.load(fh)
    rng = np.random.default_rng(5)
    seen = {}
    for name, tau, fd, edges, eta in revmap_probe.cases():
        grid = emu._Grid(tau, fd, edges)
        n = grid.M
        v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.exp(-np.linspace(-2, 2, n) ** 2)
        w = np.array([-3.7 if name == "flat" else 2.9])
        if n <= 300:
            rng.standard_normal((n, n)), rng.standard_normal((n, n))       # (the probe's explicit matrix: the same random stream)
        info = {}
        rec = emu._rev_map_dev(grid.geom, emu.to_device(grid.th_cents, torch.float64), n, eta, True, vec_t=emu.to_device(v),
                               w_t=emu.to_device(w, torch.float64), info=info).cpu().numpy()
        ref = np.nan_to_num(to.rev_map(np.outer(v, np.conj(v)) * np.abs(w[0]), tau, fd, eta, edges, True))
        assert np.abs(rec - ref).max() <= 1e-12 * np.abs(ref).max(), name
        assert np.array_equal(rec != 0, ref != 0), name
        irregular = name in ("irregular", "chunks_and_slabs_steep", "odd_axes_slabs")
        assert info["uniform_grid"] == (0 if irregular else 1), name
        digest = hashlib.sha256(np.ascontiguousarray(rec).tobytes()).hexdigest()
        if irregular:
            assert digest == general[name + "_rank1"], name
        seen[name + "_rank1"] = digest
    if os.environ.get("SCINT_WRITE_GOLDEN"):
        with open(os.path.join(here, "golden", "revmap_diag_bits.json"), "w") as fh:
            json.dump(seen, fh, indent=1)
    assert seen == pinned, [k for k in seen if pinned.get(k) != seen[k]]


def test_diagonal_back_map_fuzz_vs_oracle(emu, to):
    """Random axes (odd lengths, shifted off 0), theta grids whose step is commensurate with the Doppler step (s * step ON a column
    edge: the last bit of fl(theta_j - theta_i) decides, as in np.histogram2d), curvatures of either sign from 1 % of the arc's to
    20 times it (strided sweeps, row gathers, pairs off the delay axis): the uniform-grid kernel against the oracle."""
    import torch
    rng = np.random.default_rng(20260930)
    for trial in range(30):
        ntau, nfd = int(rng.integers(16, 1300)), int(rng.integers(8, 160))
        dt, df = 0.0137 * float(rng.uniform(0.5, 2)), 0.211 * float(rng.uniform(0.5, 2))
        tau = (np.arange(ntau) - ntau // 2) * dt
        fd = (np.arange(nfd) - nfd // 2) * df
        kind = trial % 5
        if kind == 1:
            tau, fd = tau + float(rng.uniform(-0.5, 0.5)) * dt, fd + float(rng.uniform(-0.5, 0.5)) * df
        nedge = 2 * int(rng.integers(3, 300))
        lim = float(rng.uniform(0.2, 1.1)) * fd.max() / 2
        if kind == 2:
            lim = df / int(rng.integers(1, 4)) * (nedge - 1) / 2
        if kind == 3:
            lim = df * int(rng.integers(1, 3)) * (nedge - 1) / 2
        edges = np.linspace(-lim, lim, nedge)
        eta = float(10 ** rng.uniform(-2.0, 1.3)) * (1 if rng.uniform() < 0.8 else -1) * np.abs(tau).max() / lim ** 2
        grid = emu._Grid(tau, fd, edges)
        n = grid.M
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        w = np.array([float(rng.uniform(-3, 3))])
        info = {}
        rec = emu._rev_map_dev(grid.geom, emu.to_device(grid.th_cents, torch.float64), n, eta, True, vec_t=emu.to_device(v),
                               w_t=emu.to_device(w, torch.float64), info=info).cpu().numpy()
        ref = np.nan_to_num(to.rev_map(np.outer(v, np.conj(v)) * np.abs(w[0]), tau, fd, eta, edges, True))
        key = (trial, ntau, nfd, n, eta)
        assert info["uniform_grid"] == 1, key
        assert np.abs(rec - ref).max() <= 1e-12 * np.abs(ref).max(), key
        assert np.array_equal(rec != 0, ref != 0), key


def test_results_do_not_depend_on_the_schedule(tmp_path):
    """Waves, lanes and blocks interpreted in the opposite order (SCINT_EMU_ORDER=rev) must give the
    same bits for every product of the path (FFT, gather, the two-vector Lanczos sweep, eigenvectors,
    rev_map, model, chi^2): any order is a legal GPU
    schedule, so a difference would be a missing barrier or an inter-block dependence."""
    import subprocess
    try:
        import emulated
        emulated.load()                      # build once, before the two children race for it
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:
        pytest.skip(f"host interpreter could not be built: {exc}")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "order_probe.py")
    procs = []
    for tag, order in (("fwd", ""), ("rev", "rev")):
        env = dict(os.environ, SCINT_EMU_ORDER=order, OPENBLAS_NUM_THREADS="1")
        procs.append((tag, subprocess.Popen([sys.executable, probe, str(tmp_path / f"{tag}.npz")], env=env,
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for tag, p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode(errors="replace")[-2000:]
    a, b = np.load(tmp_path / "fwd.npz"), np.load(tmp_path / "rev.npz")
    assert set(a.files) == set(b.files) and len(a.files) >= 14
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_gather_fuzz_bit_equal(emu, to):
    """Random small geometries (sizes, padding, edge counts and ranges, curvatures from far below to far
    above the arc, both Hermitian settings): the gather kernel's index arithmetic (exact floor, crop,
    wrap-around) against the oracle, bit for bit."""
    from scintools_amd.synth import arc_dynspec
    rng = np.random.default_rng(20260921)
    checked = 0
    for trial in range(40):
        nf, nt = int(rng.integers(8, 72)), int(rng.integers(8, 72))
        npad = int(rng.integers(0, 3))
        dyn, freqs, times, eta_true = arc_dynspec(nf, nt, seed=trial, nimg=5)
        fd = to.fft_axis(times, 1000.0, npad)
        tau = to.fft_axis(freqs, 1.0, npad)
        CS = to.conjugate_spectrum(dyn - dyn.mean(), npad)
        nedge = 2 * int(rng.integers(2, 40))                          # even: an odd count has two centres of equal |theta|
        lim = fd.max() * float(rng.uniform(0.2, 1.2))
        edges = np.linspace(-lim, lim, nedge)
        eta = eta_true * float(10 ** rng.uniform(-1.5, 1.5))
        hermetian = bool(trial % 2 == 0)
        try:
            ref, e_ref = to.thth_redmap(CS, tau, fd, eta, edges, hermetian)
        except (IndexError, ValueError):
            with pytest.raises((IndexError, ValueError)):
                emu.thth_redmap(CS, tau, fd, eta, edges, hermetian)
            continue
        got, e_got = emu.thth_redmap(CS, tau, fd, eta, edges, hermetian)
        assert got.shape == ref.shape, (trial, nf, nt, npad, nedge)
        assert np.array_equal(got, ref, equal_nan=True), (trial, nf, nt, npad, nedge, eta / eta_true, hermetian)
        assert np.array_equal(e_got, e_ref)
        checked += 1
    assert checked >= 25


def test_sweep_on_noise_spectra_vs_lapack(emu, to):
    """Noise-like conjugate spectra (no arc: small spectral gaps, matrices small enough that the Krylov
    space becomes complete): the two-vector sweep against LAPACK's eigvalsh of the oracle's reduced
    theta-theta."""
    rng = np.random.default_rng(5)
    for trial in range(2):
        nf, nt = int(rng.integers(100, 200)), int(rng.integers(100, 200))
        dyn = rng.standard_normal((nf, nt))
        if trial % 2:
            dyn += 5 * np.outer(np.cos(np.arange(nf) * 0.3), np.cos(np.arange(nt) * 0.2))
        fd = to.fft_axis(np.arange(nt) * 30.0, 1000.0, 0)
        tau = to.fft_axis(1400 + np.arange(nf) * 0.1, 1.0, 0)
        CS = to.conjugate_spectrum(dyn - dyn.mean(), 0)
        edges = np.linspace(-fd.max() / 2, fd.max() / 2, 2 * int(rng.integers(20, 80)))
        etas = tau.max() / (fd.max() / 2) ** 2 * np.array([0.3, 1.0, 3.0])
        ref = np.array([np.linalg.eigvalsh(to.thth_redmap(CS, tau, fd, e, edges)[0])[-1] for e in etas])
        eigs, info = emu.eval_sweep(CS, tau, fd, etas, edges, return_info=True)
        assert np.all(info["status"] == 0)
        np.testing.assert_allclose(eigs, np.abs(ref), rtol=1e-11)


def test_chisq_sweep_curvature_that_keeps_two_centres_is_nan(emu, to):
    """A curvature whose crop (ththmod.py:153-155) keeps only two theta centres has no mean edge step (ththmod.py:166:
    the mean of an empty difference): the reference's chisq_calc raises there, the sweep reports NaN for it and goes on
    with the others (as it does for the curvatures that keep fewer than two)."""
    from scintools_amd.synth import arc_dynspec
    dyn, freqs, times, eta_true = arc_dynspec(64, 48, seed=2, nimg=6)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    CS = to.conjugate_spectrum(dyn, 0)
    s = fd.max() / 2
    edges = np.array([-1.95, -0.05, 0.05, 0.15, 1.85]) * s / 2      # centres -1, 0, 0.1, 1 (x s / 2)
    th = (edges[1:] + edges[:-1]) / 2
    th = th - th[np.abs(th) == np.abs(th).min()]
    eta_all = 0.5 * np.abs(tau).max() / th[-1] ** 2                 # keeps all four centres
    eta_two = 0.5 * (np.abs(tau).max() / th[2] ** 2 + np.abs(tau).max() / th[-1] ** 2)   # keeps 0 and the small one
    eta_one = 4.0 * np.abs(tau).max() / th[2] ** 2                  # keeps the centre only
    assert [len(np.nonzero((th**2 * e < np.abs(tau.max())) * (np.abs(th) < np.abs(fd.max()) / 2))[0])
            for e in (eta_all, eta_two, eta_one)] == [4, 2, 1]
    got = emu.chisq_sweep(dyn, CS, tau, fd, np.array([eta_all, eta_two, eta_one, eta_all]), edges, 2.0)
    ref = to.chisq_calc(dyn, CS, tau, fd, eta_all, edges, 2.0)
    np.testing.assert_allclose(got[[0, 3]], [ref, ref], rtol=1e-9)
    assert np.isnan(got[1]) and np.isnan(got[2])
    with pytest.raises(Exception):
        to.chisq_calc(dyn, CS, tau, fd, eta_two, edges, 2.0)


def test_default_batch_follows_the_librarys_workgroup_count(emu):
    """the resident-curvature rule counts the mat-vec workgroups the BUILT kernel launches per matrix
    (scint_sweep_workgroups: block rows per workgroup x tiles per strip are build constants), and reproduces the slot counts
    that were measured at N = 4095 -- 107 / 90 for the float64 eigenvalue / eigenpair sweeps, 214 / 138 for the mixed ones."""
    from scintools_amd import _lib
    lib = _lib.load()
    assert lib.scint_sweep_workgroups(64, 0) == sum(-(-(64 - i) // 12) for i in range(0, 64, 8)) == 27
    assert lib.scint_sweep_workgroups(64, 1) == 27 and lib.scint_sweep_workgroups(1, 0) == 1 and lib.scint_sweep_workgroups(0, 0) < 0
    assert emu.default_batch(4095, 256) == 107 and emu.default_batch(4095, 256, eigenvalues_only=False) == 90
    assert emu.default_batch(4095, 40) == 40
    assert emu.default_batch(8191, 256) in (30, 31) and emu.default_batch(2047, 256) == 256
    prev = emu.sweep_precision("mixed")
    try:
        # (214 wanted; the 32-GiB budget for resident matrices -- a mixed slot also holds the complex64 copy -- caps it at 157)
        assert emu.default_batch(4095, 256) == 157 and emu.default_batch(4095, 256, eigenvalues_only=False) == 90
        assert emu.default_batch(2047, 1000) == 256
        emu.sweep_precision("mixed-all")
        assert emu.default_batch(4095, 256, eigenvalues_only=False) == 138
    finally:
        emu.sweep_precision(prev)


def test_eval_sweep_multi_with_device_built_crop_tables(emu, to, case):
    """Round 5: the multi-chunk sweeps (Dynspec.fit_thetatheta / thetatheta_chunks) build every chunk's crop table on the device
    (scint_sweep_keep, chunk by chunk into one tensor) instead of on the host: the curves equal the per-chunk sweeps' bit for
    bit, chunks with their own theta grids and eta ranges included, and the kept indices the eigenvector variant hands back
    are the reference's mask (ththmod.py:153-155)."""
    import torch
    c = case
    cs0 = emu.to_device(c["CS"], torch.complex128)
    edges2 = c["edges"] * 0.9
    etas2 = c["etas"][::-1] * 1.3
    stack = torch.stack([cs0, cs0])
    grids = [(c["tau"], c["fd"], c["edges"]), (c["tau"], c["fd"], edges2)]
    multi, info = emu.eval_sweep_multi(stack, grids, [c["etas"], etas2], return_info=True)
    one_a = emu.eval_sweep(cs0, c["tau"], c["fd"], c["etas"], c["edges"])
    one_b = emu.eval_sweep(cs0, c["tau"], c["fd"], etas2, edges2)
    assert np.array_equal(multi[0], one_a) and np.array_equal(multi[1], one_b)
    w, V, keeps, vinfo = emu.eigvec_sweep_multi(stack, grids, [c["etas"][:2], etas2[:1]])
    g2 = emu._Grid(c["tau"], c["fd"], edges2)
    assert np.array_equal(keeps[2], g2.keep(float(etas2[0]))) and np.array_equal(vinfo["N"], [len(k) for k in keeps])
    assert np.array_equal(keeps[0], emu._Grid(c["tau"], c["fd"], c["edges"]).keep(float(c["etas"][0])))


def test_chisq_from_the_back_map_accumulators(emu, to, monkeypatch):
    """Round 6: on symmetric axes chi^2 of a uniform-grid curvature is formed by the back-map workgroups themselves (interior
    pixels: fft2(model) = recov, the histogram being mirror-symmetric), the image is not written; column 0 / row 0 keep the
    partner formula; a pair on a bin edge whose mirrored pair is not in the mirrored pixel sends its curvature through the
    written image again.  Against SCINT_CHISQ_FUSE=0 (every image written, the Parseval pass of round 4) to 1e-12 and
    against the oracle to 1e-9, with the route the call reports: (a) a generic grid -- fused, (almost) nothing redone -- and
    curvatures that put delays exactly on row edges -- fused, some redone;
    (b) theta step = half the Doppler step, i.e. every odd diagonal ON a column edge -- fused, the curvatures redone, same
    chi^2; (c) axes shifted off 0 and (d) odd lengths -- not fused at all; and two delay slabs with bands from a few rows to the
    whole axis (row 0 in the band) in (a)."""
    from scintools_amd.synth import arc_dynspec

    def both(dyn, CS, tau, fd, etas, edges):
        monkeypatch.setenv("SCINT_CHISQ_FUSE", "1")
        a, ia = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, return_info=True)
        monkeypatch.setenv("SCINT_CHISQ_FUSE", "0")
        b, ib = emu.chisq_sweep(dyn, CS, tau, fd, etas, edges, 3.0, return_info=True)
        assert not ib["fused"] and ib["redone"] == 0
        np.testing.assert_allclose(a, b, rtol=1e-12)
        return a, ia

    dyn, freqs, times, eta_true = arc_dynspec(1100, 48, seed=21, nimg=8, noise=0.05)
    dyn = dyn - dyn.mean()
    fd, tau = to.fft_axis(times, 1000.0, 0), to.fft_axis(freqs, 1.0, 0)
    CS = to.conjugate_spectrum(dyn, 0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, 60)
    th = emu._Grid(tau, fd, edges).th_cents
    eta_full = np.abs(tau).max() / (th**2).max()
    etas = np.array([0.004, 0.03, 0.3, 0.49, 0.51, 0.95, 1.05, 1.6, 3.0, 0.2]) * eta_full
    a, info = both(dyn, CS, tau, fd, etas * 1.0123456, edges)                          # (a)
    assert info["fused"] and info["redone"] <= 1
    # (these curvatures are rational multiples of |tau|max / theta_max^2: eta (theta_j^2 - theta_i^2) ON delay edges -- four of the ten are done again)
    a, info = both(dyn, CS, tau, fd, etas, edges)
    assert info["fused"] and info["redone"] >= 2
    ref = np.array([to.chisq_calc(dyn, CS, tau, fd, etas[i], edges, 3.0) for i in (0, 3, 6, 8)])
    np.testing.assert_allclose(a[[0, 3, 6, 8]], ref, rtol=1e-9)
    step = fd[1] - fd[0]
    n = 40
    edges_c = (np.arange(n) - (n - 1) / 2) * (step / 2)                                # (b) centres at multiples of step / 2 ... the differences too
    b_, info = both(dyn, CS, tau, fd, etas[2:7], edges_c)
    assert info["fused"] and info["redone"] >= 1
    np.testing.assert_allclose(b_[1], to.chisq_calc(dyn, CS, tau, fd, etas[3], edges_c, 3.0), rtol=1e-9)
    c_, info = both(dyn, CS, tau + 0.3 * (tau[1] - tau[0]), fd, etas[2:5], edges)     # (c)
    assert not info["fused"]
    dyn2, freqs2, times2, _ = arc_dynspec(97, 81, seed=11, nimg=8, noise=0.05)          # (d)
    dyn2 = dyn2 - dyn2.mean()
    fd2, tau2 = to.fft_axis(times2, 1000.0, 0), to.fft_axis(freqs2, 1.0, 0)
    d_, info = both(dyn2, to.conjugate_spectrum(dyn2, 0), tau2, fd2, etas[2:5] * 0 + np.array([0.7, 1.0, 1.6]) * _,
                    np.linspace(-fd2.max() / 2, fd2.max() / 2, 70))
    assert not info["fused"]


def test_chunks_sharing_a_grid_object_give_the_same_retrieval(emu, to):
    """Round 6: chunk_retrieval_batch makes ONE grid object per distinct (axes, edges) and the multi-chunk sweep builds one crop
    table per distinct (grid object, curvatures), expanded on the device -- 961 chunks of a 31 x 31 mosaic have 31 of each.  Six
    chunks in two 'frequency rows' (same axes, edges and curvature within a row): the batch's wavefields are bit-identical to the
    same chunks sent one by one (a group of one has nothing to share), and eigvec_sweep_multi gives the same eigenpairs and crops
    whether the grids of a row are one object or equal copies."""
    import torch
    from scintools_amd.synth import arc_dynspec
    chunks = []
    for row in range(2):
        base = arc_dynspec(48, 40, seed=50 + row, nimg=8)
        fd = to.fft_axis(base[2], 1000.0, 1)
        edges = np.linspace(-fd.max() / 2, fd.max() / 2, 40 + 2 * row)
        for k in range(3):
            dyn = arc_dynspec(48, 40, seed=60 + 3 * row + k, nimg=8)[0]
            chunks.append((dyn - dyn.mean(), edges, base[2], base[1], base[3] * (1.0 + 0.1 * row)))
    together = emu.chunk_retrieval_batch(chunks[:3], 1, 0.0)          # one row: one grid object
    one_by_one = np.stack([emu.chunk_retrieval_batch([c], 1, 0.0)[0] for c in chunks[:3]])
    assert np.array_equal(together, one_by_one) and np.abs(together).max() > 0
    stack = torch.stack([emu.conjugate_spectrum(c[0], 1, pad_value=float(c[0].mean())) for c in chunks[:3]])
    tau, fd = to.fft_axis(chunks[0][3], 1.0, 1), to.fft_axis(chunks[0][2], 1000.0, 1)
    shared = emu._Grid(tau, fd, chunks[0][1])
    etas = [np.array([chunks[0][4]])] * 3
    wa, Va, ka, ia = emu.eigvec_sweep_multi(stack, [shared] * 3, etas)
    wb, Vb, kb, ib = emu.eigvec_sweep_multi(stack, [emu._Grid(tau, fd, chunks[0][1]) for _ in range(3)], etas)
    assert all(np.array_equal(x, y) for x, y in zip(wa, wb)) and np.array_equal(Va.cpu().numpy(), Vb.cpu().numpy())
    assert all(np.array_equal(x, y) for x, y in zip(ka, kb)) and np.array_equal(ia["N"], ib["N"])
