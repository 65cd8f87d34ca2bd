"""Host-side logic of the arc-normalisation row, checked on the CPU (no kernels run):
the spline system the host factors for scint_spline_resample, its frequency blocking, the
DeviceBacked attribute protocol, and the parabola fits against the reference's goldens."""
import numpy as np
import pytest
import torch
from scipy.interpolate import interp1d

from scintools_amd import arcfit
from scintools_amd.device import DeviceBacked


def _moments(x, y, block_rows=0, warm=0):
    """What spline_forward/backward/ends_kernel compute, in NumPy (same recurrences, same order)."""
    n = len(x)
    h, a, inv, sup, end = arcfit._spline_system(x)
    rows = n - 2
    R = rows if block_rows <= 0 else block_rows
    D = np.zeros_like(y)
    M = np.zeros_like(y)
    nblk = (rows + R - 1) // R
    for b in range(nblk):
        first, last = 1 + b * R, min(1 + b * R + R, n - 1)
        dp = np.zeros(y.shape[1])
        for i in range(max(1, first - warm), last):
            r = 6.0 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
            dp = (r - a[i] * dp) * inv[i]
            if i >= first:
                D[i] = dp
    for b in range(nblk):
        first, last = 1 + b * R, min(1 + b * R + R, n - 1)
        m = np.zeros(y.shape[1])
        for i in range(min(n - 2, last - 1 + warm), first - 1, -1):
            m = D[i] - sup[i] * m
            if i < last:
                M[i] = m
    M[0] = end[0] * M[1] + end[1] * M[2]
    M[n - 1] = end[2] * M[n - 2] + end[3] * M[n - 3]
    return h, M


def _evaluate(x, y, h, M, f):
    idx = np.clip(np.searchsorted(x, f, side="right") - 1, 0, len(x) - 2)
    hk = h[idx]
    A, B = (x[idx + 1] - f) / hk, (f - x[idx]) / hk
    return (A[:, None] * y[idx] + B[:, None] * y[idx + 1] + ((A**3 - A) * hk**2 / 6)[:, None] * M[idx] +
            ((B**3 - B) * hk**2 / 6)[:, None] * M[idx + 1])


@pytest.mark.parametrize("n,kind", [(4, "irregular"), (5, "irregular"), (33, "irregular"), (192, "uniform"),
                                    (700, "irregular")])
def test_spline_system_matches_scipy_cubic(n, kind):
    rng = np.random.default_rng(n)
    x = np.linspace(1200, 1500, n) if kind == "uniform" else np.sort(rng.uniform(1200, 1500, n))
    y = rng.standard_normal((n, 3)) * 5 + 10
    f = np.concatenate([[x[0], x[-1]], rng.uniform(x[0], x[-1], 60), x[1:-1]])
    h, M = _moments(x, y)
    ref = np.stack([interp1d(x, y[:, k], kind="cubic")(f) for k in range(3)], axis=1)
    assert np.abs(_evaluate(x, y, h, M, f) - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.parametrize("kind", ["uniform", "irregular"])
def test_spline_frequency_blocks_reproduce_the_sequential_sweep(kind):
    n = 1400
    rng = np.random.default_rng(3)
    x = np.linspace(1200, 1500, n) if kind == "uniform" else np.sort(rng.uniform(1200, 1500, n))
    h, a, inv, sup, end = arcfit._spline_system(x)
    block_rows, warm = arcfit._spline_blocks(a, inv, sup, n)
    assert block_rows == 128 and 8 <= warm <= 256
    y = rng.standard_normal((n, 2)) * 7
    _, seq = _moments(x, y)
    _, blk = _moments(x, y, block_rows, warm)
    assert np.abs(blk - seq).max() <= 1e-15 * np.abs(seq).max()
    # few channels: one block (the plain sweep)
    assert arcfit._spline_blocks(*arcfit._spline_system(x[:200])[1:4], 200) == (0, 0)


def test_spline_blocks_fall_back_on_a_hostile_axis():
    """Spacing that shrinks geometrically makes the forward recurrence contract too slowly for a
    short warm-up: the host must then ask for the single sequential block."""
    n = 1200
    x = 1000.0 + np.cumsum(0.5 * 0.985**np.arange(n))[::-1].cumsum() * 1e-3
    x = np.sort(x)
    h, a, inv, sup, end = arcfit._spline_system(x)
    block_rows, warm = arcfit._spline_blocks(a, inv, sup, n)
    lf = np.abs(a[2:n - 1] * inv[2:n - 1])
    if block_rows:     # accepted: then every window of `warm` factors really is below 1e-22
        c = np.concatenate([[0.0], np.cumsum(np.log(np.maximum(lf, 1e-300)))])
        assert np.max(c[warm:] - c[:-warm]) <= np.log(1e-22)
    else:
        assert warm == 0


def test_device_backed_attribute_protocol():
    class Holder:
        spec = DeviceBacked("spec")

    h = Holder()
    assert not Holder.spec.present(h) and not hasattr(h, "spec")
    t = torch.arange(6, dtype=torch.float64).reshape(2, 3)     # stands in for a device tensor
    Holder.spec.park(h, t)
    assert Holder.spec.present(h) and Holder.spec.shape(h) == (2, 3)
    assert Holder.spec.tensor(h) is t                           # internal consumers: no copy
    host = h.spec                                               # first host read: copied down once ...
    assert isinstance(host, np.ndarray) and np.array_equal(host, t.numpy())
    assert h.spec is host                                       # ... and from now on the host array owns the value
    assert h.__dict__[Holder.spec.slot][1] is None              # the parked tensor is dropped
    host[0, 0] = 99.0
    assert h.spec[0, 0] == 99.0
    h.spec = np.ones((4, 4))                                    # plain assignment stores a host value
    assert Holder.spec.shape(h) == (4, 4) and h.spec.sum() == 16
    del h.spec
    assert not Holder.spec.present(h)
    with pytest.raises(AttributeError):
        h.spec


def test_parabola_fits_match_reference_goldens(golden):
    """fit_arc's last step on the reference's own profile (tests/golden/arcfit.npz)."""
    g = golden("arcfit.npz")
    eta_array, spec = g["fa_eta_array"], g["fa_spec"]
    from scipy.signal import savgol_filter
    smooth = savgol_filter(spec, 5, 1)
    pk = int(np.argmin(np.abs(smooth - np.max(smooth))))
    i1 = i2 = 1
    power = smooth[pk]
    while power > smooth[pk] - 1 and pk + i1 < len(smooth) - 1:
        i1 += 1
        power = smooth[pk - i1]
    power = smooth[pk]
    while power > smooth[pk] - 0.5 and pk + i2 < len(smooth) - 1:
        i2 += 1
        power = smooth[pk + i2]
    yfit, eta, err = arcfit.fit_parabola(eta_array[pk - i1:pk + i2], spec[pk - i1:pk + i2])
    assert eta == pytest.approx(float(g["fa_betaeta"]), rel=1e-10)
    assert err / np.sqrt(2) == pytest.approx(float(g["fa_betaetaerr2"]), rel=1e-8)
    from oracle import arcfit_oracle as ao
    got = arcfit.fit_log_parabola(eta_array[pk - i1:pk + i2], spec[pk - i1:pk + i2])
    ref = ao.fit_log_parabola(eta_array[pk - i1:pk + i2], spec[pk - i1:pk + i2])
    assert got[1] == ref[1] and got[2] == ref[2] and np.array_equal(got[0], ref[0])


def test_crop_tables_by_bisection_equal_the_reference_masks():
    """ththmod._sweep_inputs finds each curvature's crop (ththmod.py:153-155) by a bisection with the reference's own
    expression and shares rows between curvatures that keep the same centres; _reduced_centres evaluates the reduced
    edges (ththmod.py:157-172, 204-205) once per distinct crop.  Both must equal the per-curvature loop, bit for bit --
    symmetric and lopsided edges, odd and even counts, curvatures that keep nothing / everything, 0, inf and NaN; a
    negative curvature or unsorted centres take the loop itself."""
    from scintools_amd import ththmod
    rng = np.random.default_rng(3)
    for nedge, lo, hi, jitter in ((100, -1.0, 1.0, 0.0), (102, -1.0, 1.0, 0.0), (65, -0.3, 1.4, 0.0), (257, -2.0, 0.7, 0.3),
                                  (40, 0.1, 1.0, 0.0), (33, -1.0, -0.2, 0.2)):
        tau = (np.arange(96) - 48) * 0.0137
        fd = (np.arange(80) - 40) * 0.0311
        edges = np.linspace(lo, hi, nedge)
        if jitter:
            edges = np.sort(edges + rng.uniform(-jitter, jitter, nedge) * (edges[1] - edges[0]))
        grid = ththmod._Grid(tau, fd, edges)
        eta0 = np.abs(tau).max() / (np.abs(fd).max() / 2) ** 2
        etas = np.concatenate((np.geomspace(0.01, 300.0, 57) * eta0, [0.0, np.inf, np.nan, 1e-300, 1e300]))
        ki, kn = ththmod._sweep_inputs(grid, etas)
        for i, e in enumerate(etas):
            ref = grid.keep(e)
            assert kn[i] == ref.shape[0], (nedge, e)
            assert np.array_equal(ki[i, :kn[i]], ref) and not ki[i, kn[i]:].any()
        th_red = ththmod._reduced_centres(grid, ki, kn)
        for i in range(len(etas)):
            n = int(kn[i])
            want = ththmod._theta_centres(grid.edges_red(ki[i, :n])) if n >= 3 else np.zeros(0)
            assert np.array_equal(th_red[i, :len(want)], want) and not th_red[i, len(want):].any()
        assert ththmod._keep_ranges(grid, np.array([1.0, -2.0])) is None           # the loop handles these
        ki2, kn2 = ththmod._sweep_inputs(grid, np.array([eta0, -eta0]))
        assert np.array_equal(ki2[1, :kn2[1]], grid.keep(-eta0))


@pytest.mark.parametrize("nf,nt,nedge,span,irregular,shift", [(512, 512, 512, (0.25, 40.0), False, 0.0), (512, 300, 401, (0.1, 300.0), True, 0.0),
                                                              (256, 256, 300, (0.5, 2000.0), False, 0.013), (256, 256, 256, (1.0, 30000.0), True, 0.0)])
def test_reduced_centres_from_ranges_are_the_reference_expression_bit_for_bit(nf, nt, nedge, span, irregular, shift):
    """ththmod._reduced_centres_of_ranges (round 5: the chi^2 sweep's host tables without the [neta, M] index table) against the
    generic path, which evaluates the reference's expressions (ththmod.py:157-172, :83-84) per crop as they stand: every row
    equal bit for bit, the same groups of identical rows -- regular, irregular and shifted grids, crops from the whole grid
    down to a single centre (an end value closer to zero than any interior one takes the generic path inside)."""
    from scintools_amd import ththmod as thth
    from scintools_amd.synth import arc_axes
    freqs, times, _, _ = arc_axes(nf, nt)
    fd, tau = thth.fft_axis(times, 1000.0), thth.fft_axis(freqs, 1.0)
    edges = np.linspace(-fd.max() / 2, fd.max() / 2, nedge) + shift
    if irregular:
        edges = np.sort(edges + np.random.default_rng(1).uniform(-0.4, 0.4, nedge) * (edges[1] - edges[0]))
    grid = thth._Grid(tau, fd, edges)
    etas = np.geomspace(span[0], span[1], 64) * 0.02
    keep_idx, keep_n = thth._sweep_inputs(grid, etas)
    ref, gref = thth._reduced_centres(grid, keep_idx, keep_n, return_groups=True)
    first, n = thth._keep_ranges(grid, etas)
    assert np.array_equal(n, keep_n)
    got, ggot = thth._reduced_centres_of_ranges(grid, first, n)
    assert np.array_equal(got, ref) and np.array_equal(ggot, gref)
    assert len(set(gref.tolist())) > 20 and keep_n.min() < keep_n.max()


@pytest.mark.parametrize("shape", [(1, 1, 8, 6), (1, 4, 8, 6), (3, 1, 10, 12), (5, 4, 32, 20)])
def test_mosaic_with_cached_masks_is_the_reference_loop_bit_for_bit(shape):
    """ththmod.mosaic (ththmod.py:1492-1554; host NumPy, sequentially dependent) builds the chunk weight once per
    combination of neighbours (nine arrays) instead of once per chunk: the same multiplications in the same order, so
    the wavefield equals the oracle's restatement of the reference loop bit for bit."""
    from oracle import thth_oracle as to
    from scintools_amd import ththmod as thth
    rng = np.random.default_rng(7)
    chunks = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    assert np.array_equal(thth.mosaic(chunks), to.mosaic(chunks))
