"""The Quantity path of the boundary (ththmod.unit_checks, ththmod.py:1639-1668) with an astropy-like
units module on the path.  Real astropy is not installable here; the stand-in under
tests/golden/refshim (the one the golden generator runs the unmodified reference with) implements
the Quantity behaviour the boundary relies on.  Runs in a child process so that the stand-in never
leaks into this interpreter."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, warnings
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import astropy.units as u
from scintools_amd import units, ththmod
assert units.HAVE_ASTROPY

times = np.arange(64) * 30.0 * u.s
freqs = (1400.0 + 0.1 * np.arange(48)) * u.MHz
# fft_axis with Quantities: the reference's own call (ththmod.py:473-493) and its plain-float twin
fd_q = ththmod.fft_axis(times, u.mHz, 1)
tau_q = ththmod.fft_axis(freqs, u.us, 1)
fd_f = ththmod.fft_axis(np.asarray(times.value), 1000.0, 1)
tau_f = ththmod.fft_axis(np.asarray(freqs.value), 1.0, 1)
assert np.array_equal(np.asarray(fd_q.value), fd_f) and np.array_equal(np.asarray(tau_q.value), tau_f)

# strip: equivalent units are converted, bare numbers are assumed (with a warning), others raise
assert np.allclose(units.strip(fd_q.to(u.Hz), "fd", "mHz"), fd_f)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    assert np.array_equal(units.strip(fd_f, "fd", "mHz"), fd_f)
    assert any("missing units" in str(x.message) for x in w)
try:
    units.strip(tau_q, "fd", "mHz")
    raise SystemExit("incompatible units were accepted")
except u.UnitConversionError:
    pass

# unit_checks mirrors the reference: dimensionless -> desired unit (warning), equivalent -> converted
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    q = ththmod.unit_checks(0.02, "eta", u.s**3)
    assert any("missing units" in str(x.message) for x in w)
assert float(q.value) == 0.02 and q.unit.is_equivalent(u.s**3)
q2 = ththmod.unit_checks(5.0 * u.Hz, "fd", u.mHz)
assert np.isclose(float(q2.value), 5000.0)
try:
    ththmod.unit_checks(1.0 * u.s, "fd", u.mHz)
    raise SystemExit("unit_checks accepted seconds for mHz")
except u.UnitConversionError:
    pass

# the grid the kernels see is the same whether the caller passes Quantities or bare numbers
edges_q = np.linspace(-1.0, 1.0, 20) * fd_q.max() / 2
g_q = ththmod._Grid(tau_q, fd_q, edges_q)
g_f = ththmod._Grid(tau_f, fd_f, np.linspace(-1.0, 1.0, 20) * fd_f.max() / 2)
for name in ("ntau", "nfd", "tau0", "dtau", "fd0", "dfd", "tau_max", "fd_max", "tau1_step", "fd1_step"):
    assert getattr(g_q.geom, name) == getattr(g_f.geom, name), name
assert np.array_equal(g_q.th_cents, g_f.th_cents)
eta = 0.02 * u.s**3
assert np.array_equal(g_q.keep(ththmod._eta_float(eta)), g_f.keep(0.02))
er = units.attach(g_q.edges_red(g_q.keep(0.02)), "mHz")
assert er.unit.is_equivalent(u.mHz)
me_q = ththmod.min_edges(fd_q.max() / 2, fd_q, tau_q, eta)
me_f = ththmod.min_edges(fd_f.max() / 2, fd_f, tau_f, 0.02)
assert np.array_equal(np.asarray(me_q.value), np.asarray(getattr(me_f, "value", me_f)))
print("quantity path ok")
'''


def test_quantity_path_with_astropy_like_units():
    shim = os.path.join(REPO, "tests", "golden", "refshim")
    out = subprocess.run([sys.executable, "-c", CHILD, shim, REPO], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "quantity path ok" in out.stdout


POOL_CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import astropy.units as u
from scintools_amd import dynspec, units
assert units.HAVE_ASTROPY


class FakePool:                 # what sweep.gpu_pool's workers hand back when astropy is importable
    def map(self, fn, pars):
        out = []
        for k, p in enumerate(pars):
            curve = np.arange(4.0) + k if k != 1 else np.arange(3.0)     # chunk 1: one curvature failed
            out.append(((0.02 + 0.001 * k) * u.s**3, 0.002 * u.s**3, 1400.0 * u.MHz, 10.0 * u.s, curve))
        return out


d = object.__new__(dynspec.Dynspec)
d.cwf = 1; d.ncf_fit = 2; d.nct_fit = 2; d.neta = 4; d.fref = 1400.0
d.freqs = 1400.0 + np.arange(8.0); d.times = 30.0 * np.arange(8.0)
d._chunk = lambda cf, ct: (slice(4 * cf, 4 * cf + 4), slice(4 * ct, 4 * ct + 4))
d._search_params = lambda cf, ct, verbose: [cf, ct]
d.fit_thetatheta(pool=FakePool())
assert d.eta_evo.dtype == np.float64 and np.allclose(d.eta_evo.ravel(), 0.02 + 0.001 * np.arange(4))
assert np.allclose(d.eta_evo_err, 0.002)
assert np.array_equal(d.thth_eigs[0, 0], np.arange(4.0)) and np.all(np.isnan(d.thth_eigs[0, 1]))
assert np.isfinite(d.ththeta) and np.isfinite(d.ththetaerr)
print("pool path ok")
'''


def test_fit_thetatheta_pool_path_strips_quantities():
    """with astropy importable single_search returns Quantities; the pool path must strip
    them before the plain-float arrays take them, and keep complete curves."""
    shim = os.path.join(REPO, "tests", "golden", "refshim")
    out = subprocess.run([sys.executable, "-c", POOL_CHILD, shim, REPO], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pool path ok" in out.stdout
