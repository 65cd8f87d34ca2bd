"""CPU-side checks of the boundary: the shared library builds/loads without a GPU and
exports exactly what include/scint_hip.h declares; argument errors come back as status
codes with a message; the product refuses to compute without a GPU."""
import ctypes

import numpy as np
import pytest

from scintools_amd import _lib


@pytest.fixture(scope="module")
def lib():
    from scintools_amd import build
    build.build(verbose=False)
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    declared = _lib.header_symbols()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib._SIGNATURES)     # the binding covers the whole header


def test_version_and_struct_layout(lib):
    assert lib.scint_version() == _lib.ABI_VERSION      # (_lib.load refuses any other build: a stale library fails at load, not in its first call)
    assert ctypes.sizeof(_lib.CsGeom) == 2 * 8 + 8 * 8


def test_workspace_queries_and_argument_errors(lib):
    n = ctypes.c_size_t()
    assert lib.scint_sspec_workspace_bytes(4096, 4096, ctypes.byref(n)) == 0
    assert n.value >= 8192 * 8192 * 16
    assert lib.scint_cs_workspace_bytes(64, 32, 3, ctypes.byref(n)) == 0
    assert n.value >= 256 * 128 * 16
    assert lib.scint_eval_sweep_workspace_bytes(511, 16, 4, 300, ctypes.byref(n)) == 0
    assert n.value >= 4 * 511 * 511 * 8      # Hermitian tile-packed: 8 N^2 per resident eta
    # bad arguments: status code + message, no exception, no crash
    assert lib.scint_sspec_workspace_bytes(0, 10, ctypes.byref(n)) == 1
    assert "bad shape" in _lib.last_error()
    assert lib.scint_fft2(None, None, 8, 16, None, 0, None) == 1
    assert "null" in _lib.last_error()
    with pytest.raises(_lib.ScintHipError):
        _lib.check(lib.scint_fft2(None, None, 8, 16, None, 0, None), "scint_fft2")
    # arc-normalisation entry points: sizes are host arithmetic, pointers are checked before any launch
    assert lib.scint_masked_colavg_workspace_bytes(100, 300, ctypes.byref(n)) == 0
    assert n.value == 8 * 3 * 4 * 300            # 4 partials of 32 rows, (sum, weight, count) per column
    assert lib.scint_norm_sspec(None, 8, 8, None, None, 0, 4, 1.0, 5.0, 4, 4, None, None, None, 16,
                                None, None, None, None) == 1
    assert "norm_sspec: null pointer" in _lib.last_error()
    assert lib.scint_spline_resample(None, 8, 8, 0, None, None, None, None, None, 0, 0, None, None, 4,
                                     None, None, 0, None) == 1
    assert "spline_resample: null pointer" in _lib.last_error()
    assert lib.scint_block_std(None, 8, 8, 0, 4, 2, 6, None, None, 0, None) == 1
    assert lib.scint_row_nanmean(None, 8, 8, 0, 4, None, 0, 0, None, None) == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scintools_amd import ththmod
    from scintools_amd.dynspec import Dynspec
    x = np.zeros((16, 16))
    with pytest.raises(_lib.ScintHipError):
        ththmod.thth_map(x + 0j, np.arange(16.0), np.arange(16.0), 1.0, np.linspace(-1, 1, 8))
    with pytest.raises(_lib.ScintHipError):
        ththmod.conjugate_spectrum(x, 0)

    class O:
        dyn, freqs, times = x, np.arange(16.0), np.arange(16.0)
    with pytest.raises(_lib.ScintHipError):
        Dynspec(dyn=O(), verbose=False).calc_sspec()


def test_host_grid_matches_oracle(golden):
    """Host-side grid construction (centres, crop, reduced edges) equals the oracle's."""
    from oracle import thth_oracle as to
    from scintools_amd.ththmod import _Grid, fft_axis, min_edges
    g = golden("thth_small.npz")
    grid = _Grid(g["tau"], g["fd"], g["edges_b"])
    assert np.array_equal(grid.th_cents, to.theta_centres(g["edges_b"]))
    for eta in g["etas"]:
        keep, th = to.reduced_keep(g["tau"], g["fd"], eta, g["edges_b"])
        k = grid.keep(eta)
        assert np.array_equal(k, np.nonzero(keep)[0])
        assert np.array_equal(grid.edges_red(k), to.reduced_edges(th[keep]))
    assert np.array_equal(fft_axis(g["times"], 1000.0, 1), g["fd"])
    assert np.array_equal(fft_axis(g["freqs"], 1.0, 1), g["tau"])
    me = min_edges(0.4 * g["fd"].max(), g["fd"], g["tau"], float(g["eta_true"]), 2)
    assert np.array_equal(np.asarray(me), g["min_edges"])


def test_prep_thetatheta_host_logic_matches_reference(golden):
    """Chunking, eta grid and edges of Dynspec.prep_thetatheta are pure host logic: they must
    equal what the reference derived for the tutorial recipe (dynspec_thth.rst:146-170)."""
    from scintools_amd.dynspec import Dynspec
    g = golden("fit_thetatheta.npz")

    class B:
        dyn, freqs, times, dt, df = g["dspec"], g["freq"], g["time"], float(g["dt"]), float(g["df"])
    d = Dynspec(dyn=B(), verbose=False)
    d.prep_thetatheta(cwf=64, edges_lim=.3, eta_min=30, eta_max=50)
    assert (d.cwf, d.cwt, d.ncf_fit, d.nct_fit) == (int(g["cwf"]), int(g["cwt"]), 16, 1)
    assert d.neta == int(g["neta"]) and d.npad == int(g["npad"]) and d.fw == float(g["fw"])
    assert d.fref == float(g["fref"])
    assert d.eta_min == float(g["eta_min"]) and d.eta_max == float(g["eta_max"])
    assert np.array_equal(d.edges, g["edges"])
    from scintools_amd._lib import ScintHipError
    with pytest.raises(ScintHipError, match="no HIP device"):
        d.prep_thetatheta(cwf=64, edges_lim=.3)            # bounds from fit_arc: device work, no CPU fallback
    with pytest.raises(AssertionError):
        d.prep_thetatheta(eta_min=30, eta_max=50, nedge=301)


def test_psrflux_io_matches_reference(golden, tmp_path):
    """Host-side text I/O: the reference's own parse of tests/golden/synthetic.dynspec
    (descending channels, a short first sub-integration) and a write/read round trip."""
    import os
    from scintools_amd.dynspec import Dynspec
    g = golden("psrflux.npz")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synthetic.dynspec")
    d = Dynspec(filename=here, verbose=False)
    assert np.array_equal(d.dyn, g["dyn"]) and np.array_equal(d.times, g["times"])
    assert np.array_equal(d.freqs, g["freqs"])
    for key in ("nchan", "nsub", "bw", "df", "freq", "dt", "tobs", "mjd"):
        assert getattr(d, key) == g[key], key
    assert len(d.header) == int(g["nheader"]) and d.nsub == 23          # first sub-integration removed
    out = str(tmp_path / "rt.dynspec")
    d.write_file(filename=out, verbose=False, note="roundtrip")
    d2 = Dynspec(filename=out, verbose=False)
    assert np.array_equal(d2.dyn, g["rt_dyn"]) and np.array_equal(d2.times, g["rt_times"])
    assert np.array_equal(d2.freqs, g["rt_freqs"]) and d2.mjd == float(g["rt_mjd"])
    # binary side-car: written on request, preferred while it is not older than the text file,
    # and giving the same object as the text parse
    import os as _os
    from scintools_amd import psrflux
    out2 = str(tmp_path / "sc.dynspec")
    d.write_file(filename=out2, verbose=False, sidecar=True)
    assert _os.path.exists(psrflux.sidecar_path(out2))
    head, table = psrflux.load_sidecar(out2)
    head_t, table_t = psrflux.read_table(out2)
    assert head == head_t and np.array_equal(table, table_t)
    d3 = Dynspec(filename=out2, verbose=False)
    assert np.array_equal(d3.dyn, d2.dyn) and np.array_equal(d3.times, d2.times) and d3.mjd == d2.mjd
    st0 = _os.stat(out2)
    _os.utime(out2, ns=(st0.st_atime_ns, st0.st_mtime_ns - 5_000_000_000))      # an OLDER text file (cp -p, untar): ignored too
    assert psrflux.load_sidecar(out2) is None
    _os.utime(out2, ns=(st0.st_atime_ns, st0.st_mtime_ns))
    assert psrflux.load_sidecar(out2) is not None                                # the very same file again
    _os.utime(out2, (_os.path.getmtime(out2) + 10,) * 2)      # text newer than the side-car: ignored
    assert psrflux.load_sidecar(out2) is None
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.dynspec"
        bad.write_text("# MJD0: 1\n0 0 0.0 1400.0 1.0 0\n0 1 0.0\n")
        psrflux.read_table(str(bad))


def test_host_grid_randomised_against_oracle():
    """Property test of the host grid logic (centres, crop, reduced edges, fft axes) over random
    axis lengths, paddings, edge limits and curvatures: always identical to the oracle."""
    from hypothesis import given, settings, strategies as st
    from oracle import thth_oracle as to
    from scintools_amd.ththmod import _Grid, fft_axis

    @settings(max_examples=60, deadline=None)
    @given(nf=st.integers(8, 96), nt=st.integers(8, 96), npad=st.integers(0, 3), nedge=st.integers(4, 80),
           lim=st.floats(0.05, 1.4), eta_scale=st.floats(0.02, 50.0), df=st.floats(0.01, 2.0),
           dt=st.floats(1.0, 60.0))
    def check(nf, nt, npad, nedge, lim, eta_scale, df, dt):
        freqs = 1300.0 + df * np.arange(nf)
        times = dt * np.arange(nt)
        fd = fft_axis(times, 1000.0, npad)
        tau = fft_axis(freqs, 1.0, npad)
        assert np.array_equal(fd, to.fft_axis(times, 1000.0, npad))
        assert np.array_equal(tau, to.fft_axis(freqs, 1.0, npad))
        edges = np.linspace(-lim * fd.max(), lim * fd.max(), 2 * (nedge // 2))
        eta = eta_scale * tau.max() / fd.max()**2
        grid = _Grid(tau, fd, edges)
        assert np.array_equal(grid.th_cents, to.theta_centres(edges))
        keep, th = to.reduced_keep(tau, fd, eta, edges)
        k = grid.keep(eta)
        assert np.array_equal(k, np.nonzero(keep)[0])
        if len(k) >= 3:
            assert np.array_equal(grid.edges_red(k), to.reduced_edges(th[keep]))

    check()
