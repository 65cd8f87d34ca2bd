"""Drop-in for the theta-theta functions of ``scintools.ththmod`` on one MI355X.

Same names, argument order and return values as the reference
(/root/reference/scintools/ththmod.py); the NumPy/SciPy bodies are replaced by
calls into libscint_hip.so:

=================  =====================  ========================================
function           reference lines        HIP entry point(s)
=================  =====================  ========================================
thth_map           ththmod.py:56-116      scint_thth_map
thth_redmap        ththmod.py:119-173     scint_thth_map (crop fused)
rev_map            ththmod.py:176-271     scint_rev_map
modeler            ththmod.py:274-327     scint_thth_map + scint_eigh_top +
                                          scint_rev_map + scint_model_from_recov
chisq_calc         ththmod.py:330-368     ... + scint_chisq
chisq_sweep        (a loop of chisq_calc) scint_chisq_sweep
Eval_calc          ththmod.py:371-401     scint_eval_sweep (one eta)
single_search      ththmod.py:715-895     scint_cs + scint_eval_sweep (+ SciPy fit)
eval_sweep         (the loop :788-799)    scint_eval_sweep
=================  =====================  ========================================

``tau, fd, eta, edges`` may be bare numbers (us, mHz, s**3, mHz) or, when astropy
is installed, Quantities -- converted exactly as ``unit_checks`` does.  ``CS`` may
be a NumPy array (uploaded on every call) or a ``torch`` complex128 CUDA tensor
(kept where it is; use :func:`to_device` once before a loop).  Results are NumPy
arrays like the reference's.  Grid construction, the crop mask and the parabola
fit stay on the host in NumPy/SciPy, in the reference's operation order.

There is no CPU fallback: without the HIP library or without a GPU every
function raises ``ScintHipError``.
"""
import ctypes
import warnings

import numpy as np
import torch
from scipy.optimize import curve_fit

from . import _lib, units
from . import device as _dv
from .device import empty, ptr, require_gpu, stream_ptr, workspace

DEFAULT_TOL = 1e-12       # Ritz-residual tolerance of the Lanczos eigen-solver
DEFAULT_MAX_ITER = 300    # Lanczos steps (ARPACK needs 30-50 restarts-equivalent mat-vecs)
DEFAULT_BATCH_BYTES = 32 << 30  # HBM budget for concurrently resident theta-theta matrices (of 288 GB)


# ----------------------------------------------------------------------------
# small host helpers (reference operation order)
# ----------------------------------------------------------------------------
def chi_par(x, A, x0, C):
    """Parabola for fitting the eigenvalue peak (ththmod.py:38-53)."""
    return A * (x - x0) ** 2 + C


def unit_checks(var, name, desired):
    """Reference-compatible unit coercion (ththmod.py:1639-1668); needs astropy
    for anything but bare numbers.  `desired` is an astropy unit."""
    if not units.HAVE_ASTROPY:
        raise _lib.ScintHipError("unit_checks with astropy units needs astropy")
    u = units.u
    var = var * u.dimensionless_unscaled
    if u.dimensionless_unscaled.is_equivalent(var.unit):
        var = var * desired
        warnings.warn(f"{name} missing units. Assuming {desired}.")
    elif desired.is_equivalent(var.unit):
        var = var.to(desired)
    else:
        raise u.UnitConversionError(f"{name} units ({var.unit}) not equivalent to {desired}")
    return var


def fft_axis(x, unit=None, pad=0):
    """Fourier-conjugate axis (ththmod.py:473-493).

    With astropy: same call as the reference (`unit` an astropy unit).  Without:
    `unit` is the scale factor of the conversion -- 1000.0 for s -> mHz, 1.0 for
    MHz -> us -- and a bare ndarray is returned.
    """
    if units.HAVE_ASTROPY and hasattr(x, "unit"):
        fx = np.fft.fftshift(np.fft.fftfreq((pad + 1) * x.shape[0], x[1] - x[0]).to_value(unit)) * unit
        return fx
    scale = 1.0 if unit is None else float(unit)
    fx = np.fft.fftfreq((pad + 1) * x.shape[0], x[1] - x[0])
    if scale != 1.0:
        fx = fx * scale
    return np.fft.fftshift(fx)


def min_edges(fd_lim, fd, tau, eta, factor=2):
    """Smallest edges array that oversamples the CS by `factor` (ththmod.py:1671-1705)."""
    fd_v = units.strip(fd, "fd", "mHz", warn=False)
    tau_v = units.strip(tau, "tau", "us", warn=False)
    eta_v = float(units.strip(eta, "eta", "s3", warn=False))
    lim = float(units.strip(fd_lim, "fD Limit", "mHz", warn=False))
    dtau_lim = (tau_v[1] - tau_v[0]) / factor
    dtau_lim /= 2 * eta_v * lim
    dfd_lim = (fd_v[1] - fd_v[0]) / factor
    npoints = (2 * lim) // (min(dfd_lim, dtau_lim))
    npoints += np.mod(npoints, 2)
    return units.attach(np.linspace(-lim, lim, int(npoints)), "mHz")


def _theta_centres(edges):
    th = (edges[1:] + edges[:-1]) / 2            # ththmod.py:83
    th -= th[np.abs(th) == np.abs(th).min()]     # ththmod.py:84
    return th


class _Grid:
    """Plain-float view of (tau, fd, edges) plus everything the kernels need."""

    def __init__(self, tau, fd, edges):
        self.tau = units.strip(tau, "tau", "us", warn=False)
        self.fd = units.strip(fd, "fd", "mHz", warn=False)
        self.edges = np.array(units.strip(edges, "edges", "mHz", warn=False), dtype=float)
        if self.tau.ndim != 1 or self.fd.ndim != 1 or self.tau.size < 2 or self.fd.size < 2:
            raise ValueError("tau and fd must be 1-D with at least two points")
        self.th_cents = _theta_centres(self.edges)
        g = _lib.CsGeom()
        g.ntau, g.nfd = self.tau.shape[0], self.fd.shape[0]
        g.tau0, g.dtau = float(self.tau[0]), float(np.diff(self.tau).mean())   # ththmod.py:90
        g.fd0, g.dfd = float(self.fd[0]), float(np.diff(self.fd).mean())       # ththmod.py:91
        g.tau_max = float(np.abs(self.tau.max()))
        g.fd_max = float(np.abs(self.fd.max()))
        g.tau1_step = float(self.tau[1] - self.tau[0])
        g.fd1_step = float(self.fd[1] - self.fd[0])
        self.geom = g
        self._th_dev = None

    @property
    def M(self):
        return self.th_cents.shape[0]

    def th_dev(self):
        if self._th_dev is None:
            self._th_dev = to_device(self.th_cents, torch.float64)
        return self._th_dev

    def keep(self, eta):
        """Crop of thth_redmap (ththmod.py:153-155) as ascending indices."""
        th = self.th_cents
        pnts = ((th**2) * eta < self.geom.tau_max) * (np.abs(th) < self.geom.fd_max / 2)
        return np.nonzero(pnts)[0].astype(np.int32)

    def edges_red(self, keep_idx):
        """edges of the reduced map (ththmod.py:157-172)."""
        c = self.th_cents[keep_idx]
        mid = (c[:-1] + c[1:]) / 2
        step = np.diff(mid).mean()
        return np.concatenate((np.array([mid[0] - step]), mid, np.array([mid[-1] + step])))


def to_device(CS, dtype=torch.complex128):
    """Upload a conjugate spectrum once; pass the result as ``CS`` to any function here."""
    return _dv.to_device(CS, dtype)


SWEEP_MAX_CS_ELEMENTS = (1 << 31) - 1    # the packed gather of the sweeps indexes the conjugate spectrum with 32 bits


def _check_sweep_cs(ntau, nfd):
    """The sweeps (eval / eigvec / chisq, single or multi, and the batched retrieval) refuse a conjugate spectrum of 2^31
    elements or more -- 32 GiB of complex128, e.g. a 16384^2 dynspec with npad = 3 -- inside the library (SCINT_E_ARG from
    run_sweep).  Say so here, before any workspace is allocated; the per-curvature entry points (thth_map,
    thth_redmap, modeler, Eval_calc through them) index with 64 bits and take such a spectrum."""
    if int(ntau) * int(nfd) > SWEEP_MAX_CS_ELEMENTS:
        raise ValueError(f"conjugate spectrum of {ntau} x {nfd} = {int(ntau) * int(nfd)} elements: the batched sweeps index it "
                         f"with 32 bits (limit {SWEEP_MAX_CS_ELEMENTS}); use a smaller npad, chunk the observation, or call "
                         "thth_redmap / modeler per curvature")


def _cs_dev(CS, grid):
    t = to_device(CS, torch.complex128)
    if tuple(t.shape) != (grid.geom.ntau, grid.geom.nfd):
        raise ValueError(f"CS shape {tuple(t.shape)} does not match (len(tau), len(fd)) = "
                         f"({grid.geom.ntau}, {grid.geom.nfd})")
    return t


def _eta_float(eta):
    return float(units.strip(eta, "eta", "s3", warn=False))


# ----------------------------------------------------------------------------
# device-level building blocks (torch tensors in, torch tensors out)
# ----------------------------------------------------------------------------
def _thth_dev(cs_t, grid, eta, keep_idx, hermetian):
    require_gpu()
    lib = _lib.load()
    n = int(keep_idx.shape[0])
    out = empty((n, n), torch.complex128)
    if n == 0:
        return out
    keep_t = to_device(keep_idx, torch.int32)
    rc = lib.scint_thth_map(ptr(cs_t), ctypes.byref(grid.geom), ptr(grid.th_dev()), grid.M,
                            ptr(keep_t), n, eta, 1 if hermetian else 0, ptr(out), stream_ptr())
    _lib.check(rc, "scint_thth_map")
    return out


def _host_checked(arr, grid, hermetian):
    """NumPy raises IndexError when the non-Hermitian gather wraps an fd index below -len(fd)
    (ththmod.py:104); the kernel marks such pixels NaN.  Checked on the host copy, and only
    when the grid can produce such an index at all (a NaN in CS itself must pass through)."""
    if not hermetian and arr.size:
        g, th = grid.geom, grid.th_cents
        lowest = np.floor(((th.min() - th.max()) - g.fd0 + g.dfd / 2) / g.dfd)
        if lowest < -g.nfd and np.isnan(arr.real).any():
            raise IndexError("theta-theta gather index out of bounds for the conjugate spectrum")
    return arr


def _eigh_top_dev(a_t, v0_t=None, want_vec=True, tol=DEFAULT_TOL, max_iter=DEFAULT_MAX_ITER):
    """Top ('LA') eigenpair of a Hermitian device matrix.  Returns (w float, V tensor|None, iters)."""
    lib = _lib.load()
    n = int(a_t.shape[0])
    if n < 2:
        raise ValueError("eigen-decomposition needs at least a 2x2 theta-theta matrix")
    need = ctypes.c_size_t()
    _lib.check(lib.scint_eigh_top_workspace_bytes(n, max_iter, ctypes.byref(need)), "eigh_top_workspace_bytes")
    ws = workspace.get(need.value)
    w_t = empty((1,), torch.float64)
    st_t = torch.zeros((2,), dtype=torch.int32, device=a_t.device)
    vec_t = empty((n,), torch.complex128) if want_vec else None
    rc = lib.scint_eigh_top(ptr(a_t), n, ptr(v0_t), tol, max_iter, ptr(w_t), ptr(vec_t),
                            ptr(st_t[0:1]), ptr(st_t[1:2]), ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_eigh_top")
    status, iters = (int(v) for v in st_t.cpu())
    if status != _lib.SCINT_OK:
        raise ArithmeticError(f"top-eigenpair iteration failed (status {status}, {iters} steps)")
    return float(w_t.cpu()[0]), vec_t, iters


def _rev_map_dev(grid_geom, th_t, n, eta, hermetian, thth_t=None, vec_t=None, w_t=None, info=None):
    """`info` (a dict, tests): receives `uniform_grid` -- word 9 of the call's scratch, 1 when the rank-1 Hermitian image was
    formed by the uniform-grid kernel (csrc/thth.hip: rev_diag_kernel), 0 when by the general one."""
    lib = _lib.load()
    recov = empty((grid_geom.ntau, grid_geom.nfd), torch.complex128)
    rank1 = thth_t is None
    scratch = empty((256,), torch.uint8)       # its own buffer: `workspace` may be live in a caller
    rc = lib.scint_rev_map(ptr(thth_t), ptr(vec_t), ptr(w_t), 1 if rank1 else 0, ptr(th_t), n,
                           ctypes.byref(grid_geom), eta, 1 if hermetian else 0, ptr(recov),
                           ptr(scratch), scratch.numel(), stream_ptr())
    _lib.check(rc, "scint_rev_map")
    if info is not None:
        info["uniform_grid"] = int(scratch.cpu().numpy().view(np.uint64)[9])
    return recov


def _model_dev(recov_t):
    lib = _lib.load()
    ntau, nfd = (int(v) for v in recov_t.shape)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_model_workspace_bytes(ntau, nfd, ctypes.byref(need)), "model_workspace_bytes")
    ws = workspace.get(need.value)
    model = empty((ntau, nfd), torch.float64)
    rc = lib.scint_model_from_recov(ptr(recov_t), ntau, nfd, ptr(model), ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_model_from_recov")
    return model


# ----------------------------------------------------------------------------
# reference API
# ----------------------------------------------------------------------------
def thth_map(CS, tau, fd, eta, edges, hermetian=True):
    """CS -> theta-theta, nearest-bin gather (ththmod.py:56-116)."""
    grid = _Grid(tau, fd, edges)
    cs_t = _cs_dev(CS, grid)
    keep = np.arange(grid.M, dtype=np.int32)
    return _host_checked(_thth_dev(cs_t, grid, _eta_float(eta), keep, hermetian).cpu().numpy(), grid, hermetian)


def thth_redmap(CS, tau, fd, eta, edges, hermetian=True):
    """theta-theta for the largest filled-in square within edges (ththmod.py:119-173).
    Returns (thth_red, edges_red); edges_red carries mHz when astropy is present."""
    grid = _Grid(tau, fd, edges)
    cs_t = _cs_dev(CS, grid)
    e = _eta_float(eta)
    keep = grid.keep(e)
    red = _host_checked(_thth_dev(cs_t, grid, e, keep, hermetian).cpu().numpy(), grid, hermetian)
    return red, units.attach(grid.edges_red(keep), "mHz")


def rev_map(thth, tau, fd, eta, edges, hermetian=True):
    """theta-theta -> CS weighted-histogram inverse map (ththmod.py:176-271)."""
    grid = _Grid(tau, fd, edges)
    e = _eta_float(eta)
    thth_t = to_device(thth, torch.complex128)
    n = grid.M
    if tuple(thth_t.shape) != (n, n):
        raise ValueError(f"thth shape {tuple(thth_t.shape)} does not match len(edges)-1 = {n}")
    recov = _rev_map_dev(grid.geom, grid.th_dev(), n, e, hermetian, thth_t=thth_t)
    return recov.cpu().numpy()


def _modeler_dev(cs_t, grid, e):
    """Device pipeline of modeler; returns tensors (thth_red, V, w float, recov, model, keep)."""
    keep = grid.keep(e)
    red_t = _thth_dev(cs_t, grid, e, keep, True)
    n = int(keep.shape[0])
    w, V_t, _ = _eigh_top_dev(red_t, None, want_vec=True)
    # rev_map runs on the REDUCED edges: their centres are re-derived from edges_red
    # exactly as the reference does (ththmod.py:204-205 applied to :157-172)
    th_red = _theta_centres(grid.edges_red(keep))
    th_red_t = to_device(th_red, torch.float64)
    w_t = to_device(np.array([w]), torch.float64)
    recov_t = _rev_map_dev(grid.geom, th_red_t, n, e, True, vec_t=V_t, w_t=w_t)
    model_t = _model_dev(recov_t)
    return red_t, V_t, w, recov_t, model_t, keep


def modeler(CS, tau, fd, eta, edges, hermetian=True):
    """theta-theta, its rank-1 model, the model CS and model dynamic spectrum
    (ththmod.py:274-327).  Returns (thth_red, thth2_red, recov, model, edges_red, w, V).
    V has an arbitrary global phase (as ARPACK's has)."""
    if not hermetian:
        # the reference's SVD branch double-indexes U[:, 0][:, 0] and raises (ththmod.py:317-320)
        raise IndexError("modeler(hermetian=False) raises in the reference (ththmod.py:317-320)")
    grid = _Grid(tau, fd, edges)
    cs_t = _cs_dev(CS, grid)
    e = _eta_float(eta)
    red_t, V_t, w, recov_t, model_t, keep = _modeler_dev(cs_t, grid, e)
    V = V_t.cpu().numpy()
    thth2_red = np.outer(V, np.conjugate(V))      # ththmod.py:312-313
    thth2_red *= np.abs(w)
    return (red_t.cpu().numpy(), thth2_red, recov_t.cpu().numpy(), model_t.cpu().numpy(),
            units.attach(grid.edges_red(keep), "mHz"), w, V)


def chisq_calc(dspec, CS, tau, fd, eta, edges, N, mask=None):
    """chi**2 of the theta-theta model dynamic spectrum (ththmod.py:330-368)."""
    lib = _lib.load()
    grid = _Grid(tau, fd, edges)
    cs_t = _cs_dev(CS, grid)
    _, _, _, _, model_t, _ = _modeler_dev(cs_t, grid, _eta_float(eta))
    d_t = to_device(dspec, torch.float64)
    nf, nt = (int(v) for v in d_t.shape)
    m_t = None if mask is None else to_device(np.asarray(mask, dtype=np.uint8), torch.uint8)
    out = empty((1,), torch.float64)
    rc = lib.scint_chisq(ptr(model_t), int(model_t.shape[1]), ptr(d_t), nf, nt, ptr(m_t), float(N),
                         ptr(out), stream_ptr())
    _lib.check(rc, "scint_chisq")
    return float(out.cpu()[0])


def default_batch(nmax, neta, eigenvalues_only=True):
    """Curvatures resident per launch: enough mat-vec workgroups to fill the 256 CUs several times over, within the HBM
    budget for the packed matrices (8 N^2 bytes each).

    The unit is the library's own: ``scint_sweep_workgroups(nb)`` mat-vec workgroups per matrix (block rows per workgroup and
    tiles per strip are build constants of the library -- ADVICE r4: the rule used to model the round-2 kernel's strips).  The
    targets are the slot counts measured on MI355X at N = 4095 (nb = 64: 27 workgroups per matrix, complex128 or complex64:
    eight block rows x <= 12 tiles each) times that unit:
      eigenvalue sweep, float64      107 slots (round 4: 69 / 92 / 100 / 108 / 116 -> +0 / +0.3 / +0.6 / +0.8 / +1.0 %; 20 -> 30 at
                                     N = 8191 +0.7 %; 69 -> 108 with npad = 3 +1.6 %)                        -> 2889 workgroups
      eigenPAIR sweeps, float64       90 slots (round 4, per-curvature tail: 836 / 832 / 827 eta/s at 69 / 100 / 128; round 5, batched  -> 2430
                                     tail: 854 / 857-861 / 873 / 870 at 56 / 69 / 90 / 107 on the chi^2 objective)
      mixed eigenvalue sweep         214 slots wanted, 157 within the HBM budget (a slot idles two of its ~19   -> 5778
                                     chunks around the certificate pass; 2400 eta/s, round 5 call 1)
      mixed-all eigenPAIR sweeps     138 slots                                                                 -> 3726"""
    nb = -(-nmax // 64)
    lib = _lib.load()
    mode = lib.scint_sweep_precision(-1)
    use32 = mode == 2 or (eigenvalues_only and mode == 1)
    wg = max(1, int(lib.scint_sweep_workgroups(nb, 1 if use32 else 0)))
    if use32:
        target = 5778 if eigenvalues_only else 3726
    else:
        target = 2889 if eigenvalues_only else 2430
    want = -(-target // wg)
    per_slot = 8 * (nb * 64) ** 2 + 1
    if use32:
        # a slot of the mixed sweeps also holds the complex64 copy and the Q history (ththmod.DEFAULT_BATCH_BYTES is a budget, not a limit)
        per_slot = per_slot * 3 // 2 + 130 * 32 * nb * 64
    cap = max(1, DEFAULT_BATCH_BYTES // per_slot)
    return int(max(1, min(neta, 256, want, cap)))


def sweep_precision(mode=None):
    """Operand precision of the iteration of the eigenvalue sweeps (``scint_sweep_precision``), per process.

    ``"f64"`` (default): every Lanczos pass streams the complex128 theta-theta.  ``"mixed"``: the passes stream a
    complex64 copy and the eigenvalue returned is the Ritz value of a certificate pass on the complex128 matrix that
    meets the same a-posteriori bound (same ``tol``, same status codes).  ``"mixed-all"``: the eigenPAIR sweeps
    (``eigvec_sweep``, ``chisq_sweep``, the batched retrieval) also iterate on the complex64 copy, and finish the vector on
    the complex128 matrix to the float64 sweep's own residual rule.  ``None`` only queries.  Returns the mode that was in
    force before the call.

    What it buys is HBM bytes: 1.6x the sweep rate at N = 4095 on one MI355X.  Sweeps of small matrices
    (the 64 x 150 chunks of the tutorial data: N ~ 100) are bound by launch latency, not bytes, and a curvature spends
    two to three extra chunks of passes around its certificate: leave those in ``"f64"``."""
    lib = _lib.load()
    codes = {None: -1, "f64": 0, "mixed": 1, "mixed-all": 2}
    if mode not in codes:
        raise ValueError("sweep_precision: mode must be 'f64', 'mixed', 'mixed-all' or None")
    return {0: "f64", 1: "mixed", 2: "mixed-all"}[lib.scint_sweep_precision(codes[mode])]


def eval_sweep(CS, tau, fd, etas, edges, tol=DEFAULT_TOL, max_iter=DEFAULT_MAX_ITER, batch=None,
               return_info=False):
    """Dominant eigenvalue for every curvature in `etas`: the loop of
    single_search (ththmod.py:788-799) as one batched device call.

    Failed curvatures are NaN, as in the reference.  With `return_info` also
    returns a dict with per-eta matrix sizes N, Lanczos steps and status."""
    lib = _lib.load()
    grid = _Grid(tau, fd, edges)
    _check_sweep_cs(grid.geom.ntau, grid.geom.nfd)
    cs_t = _cs_dev(CS, grid)
    etas_v = np.atleast_1d(units.strip(etas, "etas", "s3", warn=False)).astype(float)
    neta, M = etas_v.shape[0], grid.M
    keep_t, keep_n = _sweep_inputs_dev(grid, etas_v)
    nmax = max(int(keep_n.max()), 1)
    if batch is None:
        batch = default_batch(nmax, neta)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_eval_sweep_workspace_bytes(M, neta, batch, max_iter, ctypes.byref(need)),
               "eval_sweep_workspace_bytes")
    ws = workspace.get(need.value)
    eigs_t = empty((neta,), torch.float64)
    st_t = empty((2, neta), torch.int32)              # status / step counts: initialised inside the library
    etas_c = np.ascontiguousarray(etas_v)
    rc = lib.scint_eval_sweep(ptr(cs_t), ctypes.byref(grid.geom), ptr(grid.th_dev()), M, ptr(keep_t),
                              keep_n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                              etas_c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), neta,
                              tol, max_iter, batch, ptr(eigs_t), ptr(st_t[0]), ptr(st_t[1]),
                              ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_eval_sweep")
    eigs = eigs_t.cpu().numpy()
    st = st_t.cpu().numpy()
    eigs[st[0] != 0] = np.nan            # failures -> NaN (ththmod.py:795-799)
    if return_info:
        return eigs, {"N": keep_n.copy(), "iters": st[1].copy(), "status": st[0].copy(), "batch": batch}
    return eigs


def _keep_ranges(grid, etas_v):
    """The crop of thth_redmap (ththmod.py:153-155) as index RANGES: (first[neta], n[neta]) with keep_i =
    arange(first_i, first_i + n_i), or None when the crop need not be one run of centres (theta not sorted, a negative
    curvature).  On sorted centres ``(th**2 * eta < tau_max) * (|th| < fd_max / 2)`` is a prefix of the centres
    >= 0 and a suffix of those < 0 (every operation is monotone in its rounded operand), so both ends come from a
    bisection with the reference's own expression at the probes: 2 x 13 vector steps over the curvatures instead of
    neta masks over all M centres (the Python loop of :func:`_sweep_inputs` was 1 % of a 4096^2 / 256-eta chi^2 step)."""
    th = grid.th_cents
    M = th.shape[0]
    if M < 1 or not np.all(np.diff(th) >= 0) or np.any(etas_v < 0) or not np.all(np.isfinite(th)):
        return None
    th2, inside, tau_max = th**2, np.abs(th) < grid.geom.fd_max / 2, grid.geom.tau_max
    z = int(np.searchsorted(th, 0.0, side="left"))        # th[z:] >= 0

    def kept(i):                                            # the reference's mask at centre i[k] for curvature k
        return (th2[i] * etas_v < tau_max) * inside[i]

    def first_false(lo, hi, flip):
        """per curvature: the number of kept centres walking from `lo` towards `hi` (exclusive), ascending or descending"""
        lo = np.full(etas_v.shape, lo, dtype=np.int64)
        hi = np.full(etas_v.shape, hi, dtype=np.int64)
        while np.any(lo < hi):
            mid = np.minimum((lo + hi) // 2, hi - 1)         # (a finished curvature probes a valid centre and ignores it)
            mid = np.maximum(mid, 0)
            ok = kept(z - 1 - mid if flip else z + mid) & (lo < hi)
            lo = np.where(ok, mid + 1, lo)
            hi = np.where(ok | (lo >= hi), hi, mid)
        return lo
    npos = first_false(0, M - z, False) if M > z else np.zeros(etas_v.shape, dtype=np.int64)
    nneg = first_false(0, z, True) if z > 0 else np.zeros(etas_v.shape, dtype=np.int64)
    return (z - nneg).astype(np.int64), (nneg + npos).astype(np.int32)


def _sweep_inputs(grid, etas_v):
    """Crop of thth_redmap (ththmod.py:153-155) for every curvature: keep_idx[neta, M]
    (left-packed ascending indices) and keep_n[neta]; same element-wise arithmetic as
    ``_Grid.keep`` with the eta-independent parts hoisted.  Curvatures that keep the same run of centres
    (:func:`_keep_ranges`) share one row."""
    th = grid.th_cents
    neta, M = etas_v.shape[0], grid.M
    keep_idx = np.zeros((neta, M), dtype=np.int32)
    keep_n = np.zeros(neta, dtype=np.int32)
    rng = _keep_ranges(grid, etas_v)
    if rng is not None:
        first, keep_n = rng
        ar = np.arange(M, dtype=np.int32)
        for a, n in {(int(a), int(n)) for a, n in zip(first, keep_n) if n > 0}:
            keep_idx[(first == a) & (keep_n == n), :n] = ar[a:a + n]
        return keep_idx, keep_n
    th2, inside, tau_max = th**2, np.abs(th) < grid.geom.fd_max / 2, grid.geom.tau_max
    for i, e in enumerate(etas_v):
        k = np.nonzero((th2 * e < tau_max) * inside)[0]
        keep_n[i] = k.shape[0]
        keep_idx[i, : k.shape[0]] = k
    return keep_idx, keep_n


def _reduced_centres_of_ranges(grid, first, keep_n):
    """:func:`_reduced_centres` for crops that are index RANGES of sorted centres (:func:`_keep_ranges`), without the
    [neta, M] index table: (th_red[neta, M], group[neta]).  The reference's arithmetic per crop (ththmod.py:157-172, then :83-84)
    is  mid = (c[:-1] + c[1:]) / 2,  step = diff(mid).mean(),  e = [mid[0] - step, mid, mid[-1] + step],  t = (e[1:] + e[:-1]) / 2,
    t -= t[|t| == |t|.min()].  Every interior value of t is (mid[k-1] + mid[k]) / 2 of the FULL grid's mids -- the same two
    operands, the same rounding, whatever the crop -- so a crop's row is a slice of one array with its two end values and its
    own `step` (a mean over the crop's slice of diff(mid): the same numbers in the same order) put in, and the subtracted
    centre is the full grid's innermost one unless an end value is closer to zero (then the crop takes the generic path).
    Bit-identical to the generic path (tests/test_arcfit_host_cpu.py); 7.5 -> 1 ms for the 96 distinct crops of the headline sweep."""
    th = grid.th_cents
    M, neta = th.shape[0], first.shape[0]
    th_red = np.zeros((neta, M))
    group = np.full(neta, -1, dtype=np.int32)
    if M < 3:
        return None
    mid = (th[:-1] + th[1:]) / 2                       # mid[k] between centres k and k + 1
    dmid = np.diff(mid)
    cen = np.zeros(M)
    cen[1:M - 1] = (mid[1:] + mid[:-1]) / 2            # cen[j], 1 <= j <= M - 2: the interior value of any crop that holds j inside
    inner = np.abs(cen[1:M - 1])
    if inner.size == 0:
        return None
    j0 = 1 + int(np.argmin(inner))
    v0 = float(inner[j0 - 1])
    unique = int(np.sum(inner == v0)) == 1
    done = {}
    for i in range(neta):
        a, n = int(first[i]), int(keep_n[i])
        if n < 3:
            continue
        j = done.get((a, n))
        if j is not None:
            th_red[i, :n] = th_red[j, :n]
            group[i] = j
            continue
        m0, m1 = mid[a], mid[a + n - 2]                # mid[0], mid[-1] of the crop
        step = dmid[a:a + n - 2].mean() if n > 2 else np.nan
        t0 = (m0 + (m0 - step)) / 2                    # (e[1] + e[0]) / 2
        t1 = ((m1 + step) + m1) / 2                    # (e[n] + e[n - 1]) / 2
        if unique and a < j0 < a + n - 1 and abs(t0) > v0 and abs(t1) > v0:
            row = cen[a:a + n].copy()
            row[0], row[n - 1] = t0, t1
            row -= cen[j0]
            th_red[i, :n] = row
        else:                                          # an end value is the innermost one, or a tie: the reference's expression as it stands
            th_red[i, :n] = _theta_centres(grid.edges_red(np.arange(a, a + n)))
        done[(a, n)] = i
        group[i] = i
    return th_red, group


def _reduced_centres(grid, keep_idx, keep_n, return_groups=False):
    """th_red[neta, M]: the centres of the reduced edges, re-derived as rev_map does (ththmod.py:204-205 on :157-172),
    one evaluation per DISTINCT crop (on the bench workload 161 of 256 curvatures keep all 4095 centres).
    With `return_groups` also group[neta] int32: curvatures with the same id have the SAME row (copied, element for
    element) -- what ``scint_chisq_sweep`` needs to share the back-map's partner table between them; -1 = no row."""
    neta, M = keep_idx.shape
    th_red = np.zeros((neta, M))
    group = np.full(neta, -1, dtype=np.int32)
    done = {}
    for i in range(neta):
        n = int(keep_n[i])
        if n < 3:                                   # (two centres have no mean edge step: the reference's rev_map raises)
            continue
        key = (int(keep_idx[i, 0]), int(keep_idx[i, n - 1]), n)
        j = done.get(key)
        if j is not None and (key[1] - key[0] + 1 == n or np.array_equal(keep_idx[i, :n], keep_idx[j, :n])):
            th_red[i, :n] = th_red[j, :n]
            group[i] = j
        else:
            th_red[i, :n] = _theta_centres(grid.edges_red(keep_idx[i, :n]))
            done[key] = i
            group[i] = i
    return (th_red, group) if return_groups else th_red


def _sweep_inputs_dev(grid, etas_v):
    """The same crop tables built on the device (``scint_sweep_keep``: NumPy's expression with its roundings):
    keep_idx as a device tensor [neta, M], keep_n on the host.  For the eigenvalue sweep, whose host side needs
    only the counts -- the Python loop of :func:`_sweep_inputs` and the 4 MB upload of its table were 3 % of a
    4096^2 / 256-eta step."""
    lib = _lib.load()
    neta, M = etas_v.shape[0], grid.M
    keep_t = empty((neta, M), torch.int32)
    n_t = empty((neta,), torch.int32)
    etas_c = np.ascontiguousarray(etas_v, dtype=np.float64)
    _lib.check(lib.scint_sweep_keep(ptr(grid.th_dev()), M, etas_c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), neta,
                                    float(grid.geom.tau_max), float(grid.geom.fd_max / 2), ptr(keep_t), ptr(n_t),
                                    stream_ptr()), "scint_sweep_keep")
    return keep_t, n_t.cpu().numpy()


def eigvec_sweep(CS, tau, fd, etas, edges, tol=DEFAULT_TOL, max_iter=DEFAULT_MAX_ITER, batch=None):
    """Dominant eigenpair of the reduced theta-theta for every curvature: modeler's
    ``eigsh(thth_red, 1, which="LA")`` (ththmod.py:308) as one batched device call.

    Returns (w[neta] float64 signed, V device tensor [neta, M] complex128 -- row i holds
    N_i = len(keep_i) entries, unit norm, arbitrary phase --, info dict)."""
    lib = _lib.load()
    grid = _Grid(tau, fd, edges)
    _check_sweep_cs(grid.geom.ntau, grid.geom.nfd)
    cs_t = _cs_dev(CS, grid)
    etas_v = np.atleast_1d(units.strip(etas, "etas", "s3", warn=False)).astype(float)
    neta, M = etas_v.shape[0], grid.M
    keep_idx, keep_n = _sweep_inputs(grid, etas_v)
    nmax = max(int(keep_n.max()), 1)
    if batch is None:
        batch = default_batch(nmax, neta, eigenvalues_only=False)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_eigvec_sweep_workspace_bytes(M, neta, batch, max_iter, ctypes.byref(need)),
               "eigvec_sweep_workspace_bytes")
    ws = workspace.get(need.value)
    keep_t = _dv.to_device(keep_idx, torch.int32)
    w_t = empty((neta,), torch.float64)
    V_t = empty((neta, M), torch.complex128)          # zeroed (rows are N_i entries long) and filled inside the library
    st_t = empty((2, neta), torch.int32)              # status / step counts: initialised inside the library
    etas_c = np.ascontiguousarray(etas_v)
    rc = lib.scint_eigvec_sweep(ptr(cs_t), ctypes.byref(grid.geom), ptr(grid.th_dev()), M, ptr(keep_t),
                                keep_n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                etas_c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), neta,
                                tol, max_iter, batch, ptr(w_t), ptr(V_t), M, ptr(st_t[0]), ptr(st_t[1]),
                                ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_eigvec_sweep")
    st = st_t.cpu().numpy()
    w = w_t.cpu().numpy()
    w[st[0] != 0] = np.nan
    info = {"N": keep_n.copy(), "iters": st[1].copy(), "status": st[0].copy(), "batch": batch,
            "keep_idx": keep_idx, "w_dev": w_t}
    return w, V_t, info


def chisq_sweep(dspec, CS, tau, fd, etas, edges, N, mask=None, return_info=False, tol=DEFAULT_TOL,
                max_iter=DEFAULT_MAX_ITER, batch=None, share_walk=True):
    """chi**2 of the theta-theta model for every curvature: the loop
    ``[chisq_calc(dspec, CS, tau, fd, eta, edges, N, mask) for eta in etas]``
    (ththmod.py:330-368) as ONE device call (``scint_chisq_sweep``): the eigenpairs of all
    curvatures come from the batched Lanczos sweep, and as each curvature retires its rank-1
    back-map, inverse FFT and chi**2 reduction are chained on a side stream while the sweep goes
    on -- no per-eta round trip through Python.  ``share_walk`` (default): curvatures that keep the same theta centres
    share the partner table of their back-maps (same pairs, same sums: not a bit changes; False exists for the test of that)."""
    lib = _lib.load()
    grid = _Grid(tau, fd, edges)
    _check_sweep_cs(grid.geom.ntau, grid.geom.nfd)
    cs_t = _cs_dev(CS, grid)
    etas_v = np.ascontiguousarray(np.atleast_1d(units.strip(etas, "etas", "s3", warn=False)).astype(float))
    neta, M = etas_v.shape[0], grid.M
    # crop tables: when the crops are index ranges of sorted centres (always on the grids of the path) the [neta, M] index table is
    # built on the device and the reduced centres from the ranges alone (round 5: the host tables were 2.8 ms of a 290-ms step)
    rng = _keep_ranges(grid, etas_v)
    fast = _reduced_centres_of_ranges(grid, rng[0], rng[1]) if rng is not None else None
    if fast is not None:
        keep_t, keep_n = _sweep_inputs_dev(grid, etas_v)
        # the device table and the host bisection evaluate the same expression; th_red and crop_group below describe the host
        # ranges, the library walks the device table: the counts AND both ends of every run must agree (a run with the same
        # count but another start, or a mask that is not one run, would pair th_red with other centres -- ADVICE r5)
        same = np.array_equal(keep_n, rng[1])
        if same and neta:
            nz = keep_n > 0
            last = torch.from_numpy(np.maximum(keep_n.astype(np.int64) - 1, 0)).to(keep_t.device)
            ends = torch.stack([keep_t[:, 0], keep_t.gather(1, last[:, None])[:, 0]]).cpu().numpy()
            same = (np.array_equal(ends[0][nz], rng[0][nz]) and np.array_equal(ends[1][nz], (rng[0] + keep_n - 1)[nz]))
        if not same:
            fast = None
    if fast is not None:
        th_red, crop_group = fast
    else:
        keep_idx, keep_n = _sweep_inputs(grid, etas_v)
        th_red, crop_group = _reduced_centres(grid, keep_idx, keep_n, return_groups=True)
        keep_t = _dv.to_device(keep_idx, torch.int32)
    if batch is None:
        batch = default_batch(max(int(keep_n.max()), 1), neta, eigenvalues_only=False)
    d_t = _dv.to_device(dspec, torch.float64)
    nf, nt = (int(v) for v in d_t.shape)
    m_t = None if mask is None else _dv.to_device(np.asarray(mask, dtype=np.uint8), torch.uint8)
    if not share_walk:
        crop_group = np.full(neta, -1, dtype=np.int32)
    crop_group = np.ascontiguousarray(crop_group, dtype=np.int32)
    keep_n = np.ascontiguousarray(keep_n, dtype=np.int32)
    th_red_t = _dv.to_device(th_red, torch.float64)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_chisq_sweep_workspace_bytes(M, neta, batch, max_iter, grid.geom.ntau, grid.geom.nfd, nf, nt,
                                                     ctypes.byref(need)), "chisq_sweep_workspace_bytes")
    ws = workspace.get(need.value)
    out = empty((neta,), torch.float64)               # NaN-initialised inside the library
    w_t = empty((neta,), torch.float64)
    V_t = empty((neta, M), torch.complex128)          # zeroed (rows are N_i entries long) and filled inside the library
    st_t = empty((2, neta), torch.int32)
    rc = lib.scint_chisq_sweep(ptr(cs_t), ctypes.byref(grid.geom), ptr(grid.th_dev()), M, ptr(keep_t),
                               keep_n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                               etas_v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), neta, tol, max_iter, batch,
                               ptr(th_red_t), crop_group.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ptr(d_t), nf, nt, ptr(m_t),
                               float(N), ptr(out), ptr(w_t), ptr(V_t), M,
                               ptr(st_t[0]), ptr(st_t[1]), ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_chisq_sweep")
    st = st_t.cpu().numpy()
    chis = out.cpu().numpy()
    chis[(st[0] != 0) | (keep_n < 3)] = np.nan   # failed curvatures (the reference's loop would raise there)
    if return_info:
        w = w_t.cpu().numpy()
        w[(st[0] != 0) | (keep_n < 3)] = np.nan      # as chis: no model exists for these curvatures
        fused, redone = ctypes.c_int32(), ctypes.c_int64()
        _lib.check(lib.scint_chisq_sweep_last_route(ctypes.byref(fused), ctypes.byref(redone)), "scint_chisq_sweep_last_route")
        return chis, {"w": w, "N": keep_n.copy(), "iters": st[1].copy(), "status": st[0].copy(), "batch": batch,
                      "fused": bool(fused.value), "redone": int(redone.value)}
    return chis


def Eval_calc(CS, tau, fd, eta, edges):
    """Dominant eigenvalue of the reduced theta-theta at one curvature (ththmod.py:371-401)."""
    eigs, info = eval_sweep(CS, tau, fd, np.array([_eta_float(eta)]), edges, return_info=True)
    if info["status"][0] != 0:
        raise ArithmeticError(f"Eval_calc failed (status {int(info['status'][0])})")
    return float(eigs[0])


def _multi_keep_dev(G, etas_all, th_stack):
    """Crop tables of MANY chunks on the device: one [neta_total, M] index tensor, filled chunk by chunk by ``scint_sweep_keep``
    (each chunk has its own theta centres and axes), and the counts on the host after ONE read-back.  Round 5: the host loop
    of :func:`_sweep_inputs` over 256 chunks and the upload of its 171-MB table were a tenth of a ``fit_thetatheta`` of a 4096^2
    observation (bench.py --workload fit_thetatheta)."""
    lib = _lib.load()
    M = G[0].M
    # (round 6) chunks that share their grid OBJECT and their curvatures (the chunks of one frequency row of the phase retrieval, where
    # chunk_retrieval_batch makes one grid per row) share their table: one call per distinct (grid, curvatures), rows expanded on the device
    first_of, cls_of = {}, []
    for c, (g, et) in enumerate(zip(G, etas_all)):
        cls_of.append(first_of.setdefault((id(g), et.tobytes()), c))
    uniq = sorted(set(cls_of))
    off_u, off = {}, 0
    for c in uniq:
        off_u[c] = off
        off += int(etas_all[c].shape[0])
    keep_u = empty((max(off, 1), M), torch.int32)
    n_u = empty((max(off, 1),), torch.int32)
    for c in uniq:
        g, et = G[c], etas_all[c]
        ne = int(et.shape[0])
        if ne == 0:
            continue
        et_c = np.ascontiguousarray(et, dtype=np.float64)
        o = off_u[c]
        _lib.check(lib.scint_sweep_keep(th_stack[c].data_ptr(), M, et_c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ne,
                                        float(g.geom.tau_max), float(g.geom.fd_max / 2), keep_u[o:o + ne].data_ptr(),
                                        n_u[o:o + ne].data_ptr(), stream_ptr()), "scint_sweep_keep")
    n_host = np.ascontiguousarray(n_u.cpu().numpy(), dtype=np.int32)
    if len(uniq) == len(G):
        return keep_u[:off], n_host[:off]
    rows = np.concatenate([np.arange(off_u[cls_of[c]], off_u[cls_of[c]] + etas_all[c].shape[0], dtype=np.int64) for c in range(len(G))]) \
        if len(G) else np.zeros(0, dtype=np.int64)
    keep_t = keep_u.index_select(0, _dv.to_device(rows, torch.int64))
    return keep_t, np.ascontiguousarray(n_host[rows], dtype=np.int32)


def eval_sweep_multi(cs_stack, grids, etas_list, tol=DEFAULT_TOL, max_iter=DEFAULT_MAX_ITER, batch=None,
                     return_info=False):
    """Eigenvalue curves of MANY chunks in one batched device call -- the chunk loop of
    Dynspec.fit_thetatheta (dynspec.py:1681-1719).

    cs_stack: CUDA complex128 tensor [nchunk, ntau, nfd]; grids: one (tau, fd, edges) triple per
    chunk (all conjugate spectra one shape, all edges one length); etas_list: per-chunk eta
    arrays.  Returns a list of eigenvalue arrays (NaN where a curvature failed)."""
    lib = _lib.load()
    G = [g if isinstance(g, _Grid) else _Grid(*g) for g in grids]
    ncs = len(G)
    cs_t = _dv.to_device(cs_stack, torch.complex128)
    if cs_t.dim() != 3 or cs_t.shape[0] != ncs:
        raise ValueError("cs_stack must be [nchunk, ntau, nfd] with one spectrum per grid")
    M = G[0].M
    if any(g.M != M or (g.geom.ntau, g.geom.nfd) != tuple(cs_t.shape[1:]) for g in G):
        raise ValueError("all chunks must share the CS shape and the number of edges")
    _check_sweep_cs(G[0].geom.ntau, G[0].geom.nfd)
    etas_all, cs_index = [], []
    for c, (g, et) in enumerate(zip(G, etas_list)):
        et = np.atleast_1d(units.strip(et, "etas", "s3", warn=False)).astype(float)
        etas_all.append(et)
        cs_index.append(np.full(et.shape[0], c, dtype=np.int32))
    etas_v = np.ascontiguousarray(np.concatenate(etas_all))
    cs_idx = np.ascontiguousarray(np.concatenate(cs_index))
    neta = etas_v.shape[0]
    geoms = (_lib.CsGeom * ncs)(*[g.geom for g in G])
    th_stack = _dv.to_device(np.stack([g.th_cents for g in G]), torch.float64)
    keep_t, keep_cnt = _multi_keep_dev(G, etas_all, th_stack)
    if batch is None:
        batch = default_batch(max(int(keep_cnt.max()), 1), neta)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_eval_sweep_multi_workspace_bytes(M, neta, batch, max_iter, ncs, ctypes.byref(need)),
               "eval_sweep_multi_workspace_bytes")
    ws = workspace.get(need.value)
    eigs_t = empty((neta,), torch.float64)
    st_t = empty((2, neta), torch.int32)
    rc = lib.scint_eval_sweep_multi(ptr(cs_t), ncs, int(cs_t.shape[1] * cs_t.shape[2]),
                                    cs_idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), geoms,
                                    ptr(th_stack), M, ptr(keep_t),
                                    keep_cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                    etas_v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), neta,
                                    tol, max_iter, batch, ptr(eigs_t), ptr(st_t[0]), ptr(st_t[1]),
                                    ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_eval_sweep_multi")
    eigs = eigs_t.cpu().numpy()
    st = st_t.cpu().numpy()
    eigs[st[0] != 0] = np.nan
    bounds = np.cumsum([0] + [e.shape[0] for e in etas_all])
    out = [eigs[bounds[i]:bounds[i + 1]] for i in range(ncs)]
    if return_info:
        return out, {"N": keep_cnt, "iters": st[1], "status": st[0], "batch": batch}
    return out


def eigvec_sweep_multi(cs_stack, grids, etas_list, tol=DEFAULT_TOL, max_iter=DEFAULT_MAX_ITER, batch=None):
    """Dominant eigenPAIRS of MANY chunks in one batched device call: modeler's
    ``eigsh(thth_red, 1, which="LA")`` (ththmod.py:308) for every (chunk, eta) -- what
    Dynspec.thetatheta_chunks needs from its per-chunk modeler calls (dynspec.py:1765-1826).

    Arguments as :func:`eval_sweep_multi`.  Returns (w list of float64 arrays, V device tensor
    [n_total, M] complex128 in chunk order, keep list of index arrays per (chunk, eta), info)."""
    lib = _lib.load()
    G = [g if isinstance(g, _Grid) else _Grid(*g) for g in grids]
    ncs = len(G)
    cs_t = _dv.to_device(cs_stack, torch.complex128)
    if cs_t.dim() != 3 or cs_t.shape[0] != ncs:
        raise ValueError("cs_stack must be [nchunk, ntau, nfd] with one spectrum per grid")
    M = G[0].M
    if any(g.M != M or (g.geom.ntau, g.geom.nfd) != tuple(cs_t.shape[1:]) for g in G):
        raise ValueError("all chunks must share the CS shape and the number of edges")
    _check_sweep_cs(G[0].geom.ntau, G[0].geom.nfd)
    etas_all, cs_index = [], []
    for c, (g, et) in enumerate(zip(G, etas_list)):
        et = np.atleast_1d(units.strip(et, "etas", "s3", warn=False)).astype(float)
        etas_all.append(et)
        cs_index.append(np.full(et.shape[0], c, dtype=np.int32))
    etas_v = np.ascontiguousarray(np.concatenate(etas_all))
    cs_idx = np.ascontiguousarray(np.concatenate(cs_index))
    neta = etas_v.shape[0]
    geoms = (_lib.CsGeom * ncs)(*[g.geom for g in G])
    th_stack = _dv.to_device(np.stack([g.th_cents for g in G]), torch.float64)
    keep_t, keep_cnt = _multi_keep_dev(G, etas_all, th_stack)
    # the callers want the kept indices on the host too: from the crop RANGES where the crops are ranges (sorted centres, eta >= 0:
    # the same expression at the bisection's probes), else from the device table
    keeps, rows_host, keep_cache = [], None, {}
    for g, et in zip(G, etas_all):
        if et.shape[0] == 1:                       # (one curvature per chunk, the phase retrieval's case: the mask itself is cheaper than a bisection)
            kk = keep_cache.get((id(g), float(et[0])))
            if kk is None:
                kk = keep_cache.setdefault((id(g), float(et[0])), g.keep(float(et[0])))
            if kk.shape[0] == int(keep_cnt[len(keeps)]):
                keeps.append(kk)
                continue
        rng = _keep_ranges(g, et)
        for k in range(et.shape[0]):
            i = len(keeps)
            if rng is not None and int(rng[1][k]) == int(keep_cnt[i]):
                keeps.append(np.arange(int(rng[0][k]), int(rng[0][k]) + int(rng[1][k]), dtype=np.int32))
            else:
                if rows_host is None:
                    rows_host = keep_t.cpu().numpy()
                keeps.append(rows_host[i, : keep_cnt[i]].copy())
    if batch is None:
        batch = default_batch(max(int(keep_cnt.max()), 1), neta, eigenvalues_only=False)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_eigvec_sweep_multi_workspace_bytes(M, neta, batch, max_iter, ncs, ctypes.byref(need)),
               "eigvec_sweep_multi_workspace_bytes")
    ws = workspace.get(need.value)
    w_t = empty((neta,), torch.float64)
    V_t = empty((neta, M), torch.complex128)
    st_t = empty((2, neta), torch.int32)
    rc = lib.scint_eigvec_sweep_multi(ptr(cs_t), ncs, int(cs_t.shape[1] * cs_t.shape[2]),
                                      cs_idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), geoms,
                                      ptr(th_stack), M, ptr(keep_t),
                                      keep_cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                      etas_v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), neta,
                                      tol, max_iter, batch, ptr(w_t), ptr(V_t), M, ptr(st_t[0]), ptr(st_t[1]),
                                      ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_eigvec_sweep_multi")
    w = w_t.cpu().numpy()
    st = st_t.cpu().numpy()
    w[st[0] != 0] = np.nan
    bounds = np.cumsum([0] + [e.shape[0] for e in etas_all])
    return ([w[bounds[i]:bounds[i + 1]] for i in range(ncs)], V_t, keeps,
            {"N": keep_cnt, "iters": st[1], "status": st[0], "batch": batch, "bounds": bounds})


def conjugate_spectrum(dspec, npad, tau=None, tau_mask=0.0, coher=True, pad_value=None, out=None):
    """Device conjugate spectrum of a chunk (ththmod.py:777-787): returns a CUDA
    complex128 tensor [(npad+1)*nf, (npad+1)*nt] (written into `out` when given)."""
    lib = _lib.load()
    if pad_value is None and not isinstance(dspec, torch.Tensor):
        # the reference's own expression for the padding value, `dspec.mean()` in NumPy (ththmod.py:783), while the chunk is still
        # on the host: the same bits whether the chunk comes alone (single_search) or in a group's array (Dynspec._fit_chunks,
        # chunk_retrieval_batch), and no device mean to read back
        pad_value = float(np.ascontiguousarray(dspec, dtype=float).mean())
    d_t = to_device(dspec, torch.float64)
    nf, nt = (int(v) for v in d_t.shape)
    if pad_value is None:
        m = ctypes.c_double()
        _lib.check(lib.scint_mean(ptr(d_t), nf * nt, ctypes.byref(m), stream_ptr()), "scint_mean")
        pad_value = m.value
    R = (npad + 1) * nf
    lo = hi = 0
    if tau is not None:
        tau_v = units.strip(tau, "tau", "us", warn=False)
        sel = np.nonzero(np.abs(tau_v) < float(units.strip(tau_mask, "tauMask", "us", warn=False)))[0]
        if sel.size:
            lo, hi = int(sel[0]), int(sel[-1]) + 1
            if hi - lo != sel.size:
                raise ValueError("tau mask is not a contiguous block of delays")
    need = ctypes.c_size_t()
    _lib.check(lib.scint_cs_workspace_bytes(nf, nt, npad, ctypes.byref(need)), "cs_workspace_bytes")
    ws = workspace.get(need.value)
    cs_t = empty((R, (npad + 1) * nt), torch.complex128) if out is None else out
    if tuple(cs_t.shape) != (R, (npad + 1) * nt) or cs_t.dtype != torch.complex128 or not cs_t.is_contiguous():
        raise ValueError("conjugate_spectrum: `out` must be a contiguous complex128 tensor of the padded shape")
    rc = lib.scint_cs(ptr(d_t), nf, nt, npad, float(pad_value), lo, hi, 0 if coher else 1,
                      ptr(cs_t), ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_cs")
    return cs_t


def fit_eig_peak(etas, eigs, fw):
    """Parabola fit around the eigenvalue peak (ththmod.py:814-859), host SciPy.
    Returns (eta_fit, eta_sig, popt) -- NaN, NaN, None on failure."""
    try:
        good = np.isfinite(eigs)
        etas = etas[good]
        eigs = eigs[good]
        sel = np.abs(etas - etas[eigs == eigs.max()]) < fw * etas[eigs == eigs.max()]
        etas_fit = etas[sel]
        eigs_fit = eigs[sel]
        C = eigs_fit.max()
        x0 = etas_fit[eigs_fit == C][0]
        if x0 == etas_fit[0]:
            A = (eigs_fit[-1] - C) / ((etas_fit[-1] - x0) ** 2)
        else:
            A = (eigs_fit[0] - C) / ((etas_fit[0] - x0) ** 2)
        popt, _ = curve_fit(chi_par, etas_fit, eigs_fit, p0=np.array([A, x0, C]))
        eta_fit = popt[1]
        eta_sig = np.sqrt((eigs_fit - chi_par(etas_fit, *popt)).std() / np.abs(popt[0]))
        return eta_fit, eta_sig, popt
    except Exception:
        return np.nan, np.nan, None


def single_search(params):
    """Curvature search for one chunk (ththmod.py:715-895).

    params = [dspec2, freq, time, etas, edges, name, plot, fw, npad, coher,
    tauMask, verbose] as in the reference.  Plotting is not supported (the
    reference's plot_func is matplotlib-only); `plot=True` is ignored with a
    warning.  Returns (eta_fit, eta_sig, freq.mean(), time.mean(), eigs).
    """
    (dspec2, freq, time, etas, edges, name, plot, fw, npad, coher, tauMask, verbose) = params
    time_v = units.strip(time, "time", "s", warn=False)
    freq_v = units.strip(freq, "freq", "MHz", warn=False)
    etas_v = units.strip(etas, "etas", "s3", warn=False)
    edges_v = units.strip(edges, "edges", "mHz", warn=False)
    fd = fft_axis(time_v, 1000.0, npad)          # ththmod.py:773
    tau = fft_axis(freq_v, 1.0, npad)            # ththmod.py:774
    cs_t = conjugate_spectrum(dspec2, npad, tau, tauMask, coher)
    eigs = eval_sweep(cs_t, tau, fd, etas_v, edges_v)
    eta_fit, eta_sig, _ = fit_eig_peak(etas_v, eigs, fw)
    if plot:
        warnings.warn("scintools_amd.single_search does not plot")
    if verbose:
        print(f"Chunk completed (eta = {eta_fit} +- {eta_sig} at {freq_v.mean()})", flush=True)
    if np.isfinite(eta_fit):
        eta_fit, eta_sig = units.attach(eta_fit, "s3"), units.attach(eta_sig, "s3")
    # the reference returns the curve with failed curvatures removed (ththmod.py:817)
    return (eta_fit, eta_sig, units.attach(freq_v.mean(), "MHz"), units.attach(time_v.mean(), "s"),
            eigs[np.isfinite(eigs)])


# ----------------------------------------------------------------------------
# phase retrieval
# ----------------------------------------------------------------------------
def _ifft2_shifted_dev(x_t, scale=1.0, crop=None):
    """scale * ifft2(ifftshift(x))[:crop[0], :crop[1]] on the device (complex128)."""
    lib = _lib.load()
    rows, cols = (int(v) for v in x_t.shape)
    cr, cc = crop if crop is not None else (rows, cols)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_fft2_workspace_bytes(rows, cols, ctypes.byref(need)), "fft2_workspace_bytes")
    ws = workspace.get(need.value)
    out = empty((cr, cc), torch.complex128)
    rc = lib.scint_ifft2_shifted(ptr(x_t), rows, cols, float(scale), cr, cc, ptr(out), ptr(ws), ws.numel(),
                                 stream_ptr())
    _lib.check(rc, "scint_ifft2_shifted")
    return out


def _wavefield_from_eigpair(grid, e, keep, V, w, shape, row_t=None, th_t=None):
    """Chunk wavefield from the dominant eigenpair (ththmod.py:1457-1470): theta-theta of the E field with only
    its theta_2 = 0 row filled with conj(V) sqrt(w), non-Hermitian back-map, shifted inverse FFT cropped to the
    chunk.  Returns a device tensor [nf, nt] complex128.  `row_t` / `th_t`: that row and the centres of the reduced edges
    already on the device (:func:`chunk_retrieval_batch` uploads them for all its chunks at once: an upload from pageable
    memory per chunk blocks the host until the stream has drained)."""
    n = int(keep.shape[0])
    dev = require_gpu()
    E_t = torch.zeros((n, n), dtype=torch.complex128, device=dev)                # allocation + memset
    if row_t is None:
        row = np.conjugate(np.asarray(V)[:n]) * np.sqrt(w)                        # n numbers: host
        row_t = torch.from_numpy(row)
    E_t[n // 2].copy_(row_t)                                                      # memcpy
    if th_t is None:
        th_t = _dv.to_device(_theta_centres(grid.edges_red(keep)), torch.float64)
    recov_E = _rev_map_dev(grid.geom, th_t, n, e, False, thth_t=E_t)
    nf, nt = shape
    return _ifft2_shifted_dev(recov_E, scale=nf * nt / 4, crop=(nf, nt))


RETRIEVAL_GROUP_BYTES = 8 << 30   # device bytes of conjugate spectra stacked per retrieval group (the rule of Dynspec._fit_chunks)


def chunk_retrieval_batch(chunks, npad, tauMask, verbose=False, group_bytes=None, dev_chunks=None, dev_pads=None, out_device=False):
    """Phase retrieval of MANY chunks of one shape (Dynspec.thetatheta_chunks, dynspec.py:1765-1826): the chunks'
    conjugate spectra in one device stack, all their dominant eigenpairs in ONE batched sweep
    (:func:`eigvec_sweep_multi`), then per chunk the back-map and the inverse FFT queued without a host
    synchronisation in between; one copy back per group.

    The chunks are processed in GROUPS whose stacked conjugate spectra stay below ``group_bytes`` (default
    ``RETRIEVAL_GROUP_BYTES`` = 8 GiB, the rule ``Dynspec._fit_chunks`` uses for the fit): an observation with many
    large chunks needs nchunk x (npad+1)^2 x nf x nt x 16 B for one stack of all of them, which the chunk-by-chunk
    reference path never did.  Tutorial-sized data is one group, as before.

    chunks: list of (dspec2[nf, nt], edges, time, freq, eta).  Returns complex [nchunk, nf, nt].  A chunk whose
    preparation, eigen-solve or back-map fails is zero and the error is printed -- what the reference's
    single_chunk_retrieval does (ththmod.py:1471-1475) -- and the other chunks are unaffected.

    ``dev_chunks`` / ``dev_pads`` (round 6): the chunks' pixels already in HBM ([nchunk, nf, nt] float64, cut there by
    :func:`chunk_cut_device`; the list's dspec2 entries are then ignored) and their padding values (host array);
    ``out_device``: return the device tensor instead of copying a gigabyte of chunks to the host (the mosaic reads them there)."""
    nf, nt = (int(v) for v in dev_chunks.shape[1:]) if dev_chunks is not None else np.asarray(chunks[0][0]).shape
    R, C = (npad + 1) * nf, (npad + 1) * nt
    per_group = max(1, int((RETRIEVAL_GROUP_BYTES if group_bytes is None else group_bytes) // (16 * R * C)))
    dev = require_gpu()
    out = torch.zeros((len(chunks), nf, nt), dtype=torch.complex128, device=dev) if out_device else np.zeros((len(chunks), nf, nt), dtype=complex)
    for g0 in range(0, len(chunks), per_group):
        group = chunks[g0:g0 + per_group]
        stack = torch.empty((len(group), R, C), dtype=torch.complex128, device=dev)
        # Round 5: nothing is uploaded chunk by chunk any more.  An upload from pageable host memory blocks the host until the
        # stream has drained, so the three per chunk (the chunk's pixels, the theta-theta row, the reduced centres) made host and GPU
        # take turns (bench.py --workload wavefield: 1.4 ms per chunk with the mat-vec busy 4 % of the time).  The group's chunks
        # travel in ONE array each way; the padding value of a chunk (its mean, ththmod.py:783) is taken on the host.
        grids, etas, live, pads = [], [], [], []
        d_all = np.empty((len(group), nf, nt)) if dev_chunks is None else None
        # (round 6) the chunks of one frequency row share their axes and edges: one _Grid per distinct (time step, frequency step,
        # lengths, edges) instead of one per chunk -- 961 grid objects were 45 ms of a 0.23-s calc_wavefield, before any GPU work is
        # queued.  The objects are read-only here (axes, centres, geometry).
        grid_cache = {}
        for k, (dspec2, edges, time, freq, eta) in enumerate(group):
            try:
                time_v = np.asarray(units.strip(time, "time2", "s", warn=False))
                freq_v = np.asarray(units.strip(freq, "freq2", "MHz", warn=False))
                edges_v = np.asarray(units.strip(edges, "edges", "mHz", warn=False))
                # (fft_axis reads only the length and the first step of an axis)
                key = (time_v.shape[0], float(time_v[1] - time_v[0]), freq_v.shape[0], float(freq_v[1] - freq_v[0]), edges_v.tobytes())
                grid = grid_cache.get(key)
                if grid is None:
                    grid = _Grid(fft_axis(freq_v, 1.0, npad), fft_axis(time_v, 1000.0, npad), edges_v)
                    grid_cache[key] = grid
                e = np.array([_eta_float(eta)])
                if (grid.geom.ntau, grid.geom.nfd) != (R, C) or (grids and grid.M != grids[0].M):
                    raise ValueError("axes or edges of this chunk do not match the chunk shape (%d, %d)" % (nf, nt))
                if dev_chunks is None:
                    d_all[len(live)] = np.asarray(dspec2, dtype=float)
                    pads.append(float(np.asarray(dspec2, dtype=float).mean()))     # (the chunk's own memory order, as dspec.mean() of ththmod.py:783)
                else:
                    pads.append(float(dev_pads[g0 + k]))
            except Exception as exc:          # this chunk stays zero; the slot of the stack is reused by the next one
                print("Chunk %d: %s" % (g0 + k, exc), flush=True)
                continue
            grids.append(grid)
            etas.append(e)
            live.append(k)
        if not live:
            continue
        # (round 6) one call for the group's conjugate spectra, one for its back-maps and inverse transforms: a dozen Python-issued
        # launches per chunk were 0.17 s of a 0.31-s calc_wavefield of 961 chunks
        lib = _lib.load()
        if dev_chunks is None:
            d_t = _dv.to_device(d_all[: len(live)], torch.float64)
        elif live == list(range(len(group))):
            d_t = dev_chunks[g0:g0 + len(group)]
        else:
            d_t = dev_chunks[g0 + torch.as_tensor(np.asarray(live), device=dev)].contiguous()
        lohi = np.zeros((len(live), 2), dtype=np.int64)
        mask_us = float(units.strip(tauMask, "tauMask", "us", warn=False))
        for j in range(len(live)):
            sel = np.nonzero(np.abs(grids[j].tau) < mask_us)[0]
            if sel.size:
                lohi[j] = (int(sel[0]), int(sel[-1]) + 1)
                if lohi[j, 1] - lohi[j, 0] != sel.size:
                    raise ValueError("tau mask is not a contiguous block of delays")
        pads_a = np.ascontiguousarray(pads, dtype=np.float64)
        need = ctypes.c_size_t()
        _lib.check(lib.scint_cs_workspace_bytes(nf, nt, npad, ctypes.byref(need)), "cs_workspace_bytes")
        ws = workspace.get(need.value)
        _lib.check(lib.scint_cs_batch(ptr(d_t), len(live), nf, nt, npad, pads_a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      lohi.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 0, ptr(stack), ptr(ws), ws.numel(), stream_ptr()),
                   "scint_cs_batch")
        w_list, V_t, keeps, info = eigvec_sweep_multi(stack[: len(live)], grids, etas)
        V = V_t.cpu().numpy()
        M = grids[0].M
        rows_all, th_all = np.zeros((len(live), M), dtype=complex), np.zeros((len(live), M))
        keep_n = np.zeros(len(live), dtype=np.int32)
        th_cache = {}                    # reduced centres per (grid, crop): shared by the chunks of a frequency row
        for j, k in enumerate(live):
            n = int(keeps[j].shape[0])
            if info["status"][j] != 0 or n < 2:
                print("Chunk %d: eigen-decomposition failed" % (g0 + k), flush=True)
                continue
            try:
                rows_all[j, :n] = np.conjugate(V[j][:n]) * np.sqrt(float(w_list[j][0]))      # ththmod.py:1459-1461
                tkey = (id(grids[j]), int(keeps[j][0]), n) if (n == int(keeps[j][-1]) - int(keeps[j][0]) + 1) else None
                th_red = th_cache.get(tkey) if tkey is not None else None
                if th_red is None:
                    th_red = _theta_centres(grids[j].edges_red(keeps[j]))
                    if tkey is not None:
                        th_cache[tkey] = th_red
                th_all[j, :n] = th_red
                keep_n[j] = n
            except Exception as exc:
                print("Chunk %d: %s" % (g0 + k, exc), flush=True)
        rows_t, th_all_t = _dv.to_device(rows_all, torch.complex128), _dv.to_device(th_all, torch.float64)
        out_t = torch.zeros((len(live), nf, nt), dtype=torch.complex128, device=dev)
        geoms = (_lib.CsGeom * len(live))(*[g_.geom for g_ in grids])
        etas_a = np.ascontiguousarray([float(e_[0]) for e_ in etas], dtype=np.float64)
        # classes: consecutive chunks with the same reduced centres, curvature and axes (the chunks of one frequency row) share the
        # pair counts of their back-maps
        class_id, prev, cid = np.zeros(len(live), dtype=np.int32), None, -1
        for j in range(len(live)):
            key = (float(etas_a[j]), int(keep_n[j]), th_all[j, :keep_n[j]].tobytes(), bytes(grids[j].geom))
            if key != prev:
                cid, prev = cid + 1, key
            class_id[j] = cid
        _lib.check(lib.scint_retrieval_tail_workspace_bytes(M, R, C, ctypes.byref(need)), "retrieval_tail_workspace_bytes")
        ws = workspace.get(need.value)
        _lib.check(lib.scint_retrieval_tail(ptr(rows_t), ptr(th_all_t), keep_n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                            class_id.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), geoms,
                                            etas_a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(live), M, nf, nt, nf * nt / 4,
                                            ptr(out_t), ptr(ws), ws.numel(), stream_ptr()), "scint_retrieval_tail")
        if verbose:
            for j, k in enumerate(live):
                if keep_n[j] >= 2:
                    print("Chunk %d success" % (g0 + k), flush=True)
        if out_device:
            out[g0 + torch.as_tensor(np.asarray(live), device=dev)] = out_t
        else:
            out[g0 + np.asarray(live)] = out_t.cpu().numpy()
        # release this group's device buffers BEFORE the next group allocates its own: while the names are bound the caching
        # allocator cannot reuse the blocks and the peak would be two groups -- the bound is `group_bytes`, not twice it
        del stack, V_t, out_t, d_t, rows_t, th_all_t, ws
    return out


def single_chunk_retrieval(params):
    """Phase retrieval on one time/frequency chunk (ththmod.py:1390-1476).

    params = (dspec2, edges, time, freq, eta, idx_t, idx_f, npad, tauMask, verbose) as in the
    reference.  Returns (model_E[nf, nt] complex, idx_f, idx_t); on failure a zero array, as
    the reference does.  The wavefield carries the arbitrary global phase of the eigenvector."""
    dspec2, edges, time, freq, eta, idx_t, idx_f, npad, tauMask, verbose = params
    time_v = units.strip(time, "time2", "s", warn=False)
    freq_v = units.strip(freq, "freq2", "MHz", warn=False)
    e = _eta_float(eta)
    edges_v = units.strip(edges, "edges", "mHz", warn=False)
    dspec2 = np.asarray(dspec2, dtype=float)
    fd = fft_axis(time_v, 1000.0, npad)
    tau = fft_axis(freq_v, 1.0, npad)
    try:
        cs_t = conjugate_spectrum(dspec2, npad, tau, tauMask, True)
        grid = _Grid(tau, fd, edges_v)
        # of modeler's outputs (ththmod.py:1455) only the eigenpair is used here: no rank-1 back-map, no model
        keep = grid.keep(e)
        w, V_t, _ = _eigh_top_dev(_thth_dev(cs_t, grid, e, keep, True), None, want_vec=True)
        model_E = _wavefield_from_eigpair(grid, e, keep, V_t.cpu().numpy(), w, dspec2.shape).cpu().numpy()
        if verbose:
            print("Chunk %s-%s success" % (idx_f, idx_t), flush=True)
    except Exception as exc:   # the reference keeps going with a zero chunk
        print(exc, flush=True)
        model_E = np.zeros(dspec2.shape, dtype=complex)
    return (model_E, idx_f, idx_t)


def mask_func(w):
    """Mask for combining chunks (ththmod.py:1479-1489)."""
    x = np.linspace(0, w - 1, w)
    return np.sin((np.pi / 2) * x / w) ** 2


def mosaic(chunks):
    """Stack half-overlapping wavefield chunks after removing their relative phase
    (ththmod.py:1492-1554).  Host NumPy: one pass over the output, sequentially dependent."""
    nct, ncf, cwf, cwt = chunks.shape[1], chunks.shape[0], chunks.shape[2], chunks.shape[3]
    E_recov = np.zeros(((ncf - 1) * (cwf // 2) + cwf, (nct - 1) * (cwt // 2) + cwt), dtype=complex)
    masks = {}          # the weight of a chunk depends only on which of its four sides have a neighbour: nine arrays, not one per chunk

    def mask_of(cf, ct):
        key = (cf > 0, cf < ncf - 1, ct > 0, ct < nct - 1)
        if key not in masks:
            mask = np.ones((cwf, cwt))
            if key[0]:
                mask[: cwf // 2, :] *= mask_func(cwf // 2)[:, np.newaxis]
            if key[1]:
                mask[cwf // 2:, :] *= 1 - mask_func(cwf // 2)[:, np.newaxis]
            if key[2]:
                mask[:, : cwt // 2] *= mask_func(cwt // 2)
            if key[3]:
                mask[:, cwt // 2:] *= 1 - mask_func(cwt // 2)
            masks[key] = mask
        return masks[key]
    for cf in range(ncf):
        for ct in range(nct):
            chunk_new = chunks[cf, ct, :, :]
            sl = (slice(cf * cwf // 2, cf * cwf // 2 + cwf), slice(ct * cwt // 2, ct * cwt // 2 + cwt))
            chunk_old = E_recov[sl]
            mask = mask_of(cf, ct)
            rot = np.angle((chunk_old * np.conjugate(chunk_new) * mask).mean())
            E_recov[sl] += chunk_new * mask * np.exp(1j * rot)
    return E_recov


_NUMPY_MODES = {}


def _numpy_mosaic_modes(cwf, cwt):
    """How THIS host's NumPy evaluates ``chunk_old * np.conjugate(chunk_new) * mask`` (ththmod.py:1550) for chunks of this
    shape -- two properties of the host, measured on random data with the very expression, so that the device mosaic can equal
    the host loop of :func:`mosaic` bit for bit (csrc/mosaic.hip: cmul_np):

    * fused: NumPy's x86 SIMD loops (AVX2 / AVX-512 with FMA3) multiply complex128 arrays as re = fma(ar, br, -(ai bi)),
      im = fma(ar, bi, ai br) -- one rounding fewer than the plain expressions of its scalar loop;
    * swapped: for temporaries of 256 KiB and more NumPy reuses ``conj(chunk_new)`` as the output of the product and, the
      operation being commutative, computes ``conj(chunk_new) * chunk_old`` -- the fused form is not symmetric in its operands.

    Returns (fused, swapped), or None if neither reading reproduces the expression (an unknown NumPy: the caller then warns
    and the device result agrees with the host loop to rounding only)."""
    key = (cwf, cwt)
    if key not in _NUMPY_MODES:
        rng = np.random.default_rng(12345)
        big = rng.standard_normal((cwf + 2, cwt + 2)) + 1j * rng.standard_normal((cwf + 2, cwt + 2))
        a = big[1:1 + cwf, 1:1 + cwt]                                 # a window of a larger array, as E_recov[sl] is
        b = rng.standard_normal((cwf, cwt)) + 1j * rng.standard_normal((cwf, cwt))
        m = rng.random((cwf, cwt))
        expr = a * np.conjugate(b) * m
        cb = np.conjugate(b)

        def plain(x, y):
            return (x.real * y.real - x.imag * y.imag) + 1j * (x.real * y.imag + x.imag * y.real)
        fused = not np.array_equal(np.multiply(a, cb), plain(a, cb))
        if np.array_equal(expr, np.multiply(np.multiply(a, cb), m)):
            _NUMPY_MODES[key] = (fused, False)
        elif np.array_equal(expr, np.multiply(np.multiply(cb, a), m)):
            _NUMPY_MODES[key] = (fused, True)
        else:
            _NUMPY_MODES[key] = None
    return _NUMPY_MODES[key]


def _mosaic_tapers(ncf, nct, cwf, cwt):
    """The taper of chunk (cf, ct) as mask[r, c] = fr[r] * fc[c] (ththmod.py:1526-1546 builds it by multiplying a matrix of ones
    by a row factor, then by a column factor: one product per element, the same bits).  Returns (rows[4, cwf], cols[4, cwt]) and
    the index function: variant 2 * (has a neighbour before) + (has a neighbour after)."""
    def variants(w):
        v = np.ones((4, w))
        mf = mask_func(w // 2)
        for k in range(4):
            if k & 2:
                v[k, : w // 2] *= mf
            if k & 1:
                v[k, w // 2:] *= 1 - mf
        return v
    return variants(cwf), variants(cwt)


def mosaic_device(chunks_t):
    """:func:`mosaic` with the chunks and the wavefield in HBM (device tensor [ncf, nct, cwf, cwt] complex128 -> device tensor
    [F, T]).  The reference's loop (ththmod.py:1548-1553) is sequential, but chunk (cf, ct) only meets its four predecessors
    (cf, ct-1) and (cf-1, ct-1 .. ct+1): chunks with equal 2 cf + ct are independent, their windows disjoint, and every
    overlapping pair keeps its order -- so the mosaic walks 2 (ncf - 1) + nct STEPS (91 for the 961 chunks of a 4096^2
    observation).  Per step one kernel forms ``(chunk_old * conj(chunk_new) * mask)`` and its sum in NumPy's own summation order
    for every chunk of the step, the sums (16 bytes each) come to the host where ``mean``, ``numpy.angle`` and ``numpy.exp``
    are NumPy's, and a second kernel adds ``chunk_new * mask * exp(1j * rot)``.  Bit-identical to the host loop on the same
    chunks (tests); the host loop spent 0.26 s on those 961 chunks and needed them on the host (1 GB)."""
    lib = _lib.load()
    ncf, nct, cwf, cwt = (int(v) for v in chunks_t.shape)
    if cwf % 2 or cwt % 2:
        raise ValueError("mosaic: chunk sizes must be even (the reference's half-overlap tapers)")
    F, T = (ncf - 1) * (cwf // 2) + cwf, (nct - 1) * (cwt // 2) + cwt
    rows, cols = _mosaic_tapers(ncf, nct, cwf, cwt)
    rows_t, cols_t = _dv.to_device(rows, torch.float64), _dv.to_device(cols, torch.float64)
    E_t = torch.zeros((F, T), dtype=torch.complex128, device=chunks_t.device)
    modes = _numpy_mosaic_modes(cwf, cwt)
    if modes is None:
        warnings.warn("scintools_amd: this NumPy evaluates the mosaic's products in an unknown way; the device mosaic agrees with "
                      "the reference's host loop to rounding, not bit for bit")
        modes = (False, False)
    fused = (1 if modes[0] else 0) | (2 if modes[1] else 0)
    # the jobs in step order, at most 64 per launch
    steps = {}
    for cf in range(ncf):
        for ct in range(nct):
            steps.setdefault(2 * cf + ct, []).append((cf, ct))
    launches, table = [], []
    for t in sorted(steps):
        for k0 in range(0, len(steps[t]), 64):
            part = steps[t][k0:k0 + 64]
            launches.append((len(table), len(part)))
            for cf, ct in part:
                table.append(((cf * (cwf // 2)) * T + ct * (cwt // 2), cf * nct + ct, 2 * (cf > 0) + (cf < ncf - 1), 2 * (ct > 0) + (ct < nct - 1)))
    jobs_t = _dv.to_device(np.asarray(table, dtype=np.int64), torch.int64)
    most = max(n for _, n in launches)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_mosaic_workspace_bytes(cwf, cwt, most, ctypes.byref(need)), "mosaic_workspace_bytes")
    ws = workspace.get(need.value)
    sums_t = empty((most, 2), torch.float64)
    count = cwf * cwt
    st = stream_ptr()
    chunks_c = chunks_t.contiguous()
    ph = np.empty((most, 2))
    for first, n in launches:
        jp = jobs_t[first].data_ptr()
        _lib.check(lib.scint_mosaic_phase(ptr(E_t), T, ptr(chunks_c), cwf, cwt, jp, n, ptr(rows_t), ptr(cols_t), fused,
                                          ptr(ws), ws.numel(), ptr(sums_t), st), "scint_mosaic_phase")
        sums = sums_t[:n].cpu().numpy()
        for k in range(n):
            tot = np.complex128(complex(sums[k, 0], sums[k, 1]))
            mean = tot.dtype.type(tot / count)                       # numpy's _mean: umr_sum(...) / rcount
            e = np.exp(1j * np.angle(mean))
            ph[k, 0], ph[k, 1] = e.real, e.imag
        _lib.check(lib.scint_mosaic_add(ptr(E_t), T, ptr(chunks_c), cwf, cwt, jp, n, ptr(rows_t), ptr(cols_t), fused,
                                        ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), st), "scint_mosaic_add")
    return E_t


def chunk_cut_device(dyn_t, origins, cwf, cwt, fortran_order=False):
    """The chunks of Dynspec.thetatheta_chunks cut on the device (dynspec.py:1782-1790: ``dspec2 = dyn[fs, ts]; dspec2 -=
    nanmean(dspec2); dspec2 = nan_to_num(dspec2)``) and the padding value of each (its mean, ththmod.py:783), NumPy's summation
    order restated (csrc/mosaic.hip).  dyn_t: device [nf, nt] float64; origins: [(r0, c0)].  Returns (chunks [n, cwf, cwt], pads [n]).
    ``fortran_order``: the host array the reference would slice is Fortran-ordered (a transposed view: how psrflux files load) --
    ``np.copy`` keeps that order and NumPy's sums walk the window column by column; the means are then formed in that order."""
    lib = _lib.load()
    nf, nt = (int(v) for v in dyn_t.shape)
    n = len(origins)
    o = np.asarray(origins, dtype=np.int32).reshape(n, 2)
    if n < 1 or o.min() < 0 or (o[:, 0] + cwf).max() > nf or (o[:, 1] + cwt).max() > nt:
        raise ValueError("chunk_cut_device: a window lies outside the dynamic spectrum")
    r0_t, c0_t = _dv.to_device(np.ascontiguousarray(o[:, 0]), torch.int32), _dv.to_device(np.ascontiguousarray(o[:, 1]), torch.int32)
    out_t, pad_t = empty((n, cwf, cwt), torch.float64), empty((n,), torch.float64)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_mosaic_workspace_bytes(cwf, cwt, n, ctypes.byref(need)), "mosaic_workspace_bytes")
    ws = workspace.get(need.value)
    _lib.check(lib.scint_chunk_cut(ptr(dyn_t), nf, nt, ptr(r0_t), ptr(c0_t), n, cwf, cwt, 1 if fortran_order else 0, ptr(out_t), ptr(pad_t), ptr(ws), ws.numel(),
                                   stream_ptr()), "scint_chunk_cut")
    return out_t, pad_t


def gerchberg_saxton_device(wavefield, dyn, tau, niter=1):
    """Gerchberg-Saxton clean-up of a wavefield (Dynspec.gerchberg_saxton, dynspec.py:1858-1875):
    host normalisation + first amplitude projection, then `niter` device iterations of
    fft2 -> zero tau < 0 -> ifft2 -> amplitude projection.  Returns the new wavefield (NumPy)."""
    lib = _lib.load()
    wf = np.array(wavefield, dtype=complex)
    F, T = wf.shape
    d = np.asarray(dyn, dtype=float)[:F, :T]
    posdspec = np.isfinite(d) * (d > 0)
    wf *= np.sqrt(d[posdspec].mean() / np.abs(wf[posdspec] ** 2).mean())
    wf[posdspec] = np.sqrt(d[posdspec]) * np.exp(1j * np.angle(wf[posdspec]))
    if niter <= 0:
        return wf
    tau_v = units.strip(tau, "tau", "us", warn=False)
    neg = np.nonzero(tau_v < 0)[0]                      # rows of the fftshifted CWF to clear
    nat = np.sort((neg - F // 2) % F)                    # the same rows in natural FFT order
    if nat.size and (nat[-1] - nat[0] + 1 != nat.size):
        raise ValueError("tau < 0 is not a contiguous block of delays")
    lo, hi = (int(nat[0]), int(nat[-1]) + 1) if nat.size else (0, 0)
    amp = np.zeros((F, T))
    amp[posdspec] = np.sqrt(d[posdspec])
    wf_t = _dv.to_device(wf, torch.complex128)
    amp_t = _dv.to_device(amp, torch.float64)
    pos_t = _dv.to_device(posdspec.astype(np.uint8), torch.uint8)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_gs_workspace_bytes(F, T, ctypes.byref(need)), "gs_workspace_bytes")
    ws = workspace.get(need.value)
    rc = lib.scint_gerchberg_saxton(ptr(wf_t), F, T, ptr(amp_t), ptr(pos_t), lo, hi, int(niter), ptr(ws),
                                    ws.numel(), stream_ptr())
    _lib.check(rc, "scint_gerchberg_saxton")
    return wf_t.cpu().numpy()


def calc_asymmetry(params):
    """Arc asymmetry of one chunk from the dominant theta-theta eigenvector
    (ththmod.py:2385-2463): (sum|V_left|^2 - sum|V_right|^2) / (sum|V_left|^2 + sum|V_right|^2).

    params = (dspec2, edges, time, freq, eta, idx_t, idx_f, npad, verbose).  Only the
    eigenvector of the reduced theta-theta is needed, so the model/back-map steps of the
    reference's modeler call are skipped.  Returns (asymm, idx_f, idx_t); NaN on failure."""
    dspec2, edges, time, freq, eta, idx_t, idx_f, npad, verbose = params
    time_v = units.strip(time, "time2", "s", warn=False)
    freq_v = units.strip(freq, "freq2", "MHz", warn=False)
    e = _eta_float(eta)
    edges_v = units.strip(edges, "edges", "mHz", warn=False)
    fd = fft_axis(time_v, 1000.0, npad)
    tau = fft_axis(freq_v, 1.0, npad)
    try:
        cs_t = conjugate_spectrum(np.asarray(dspec2, dtype=float), npad)
        w, V_t, info = eigvec_sweep(cs_t, tau, fd, np.array([e]), edges_v)
        if info["status"][0] != 0:
            raise ArithmeticError(f"eigenpair iteration failed (status {int(info['status'][0])})")
        n = int(info["N"][0])
        p = np.abs(V_t[0, :n].cpu().numpy()) ** 2
        half = (n - 1) // 2                     # cents.shape[0] == n  (ththmod.py:2447-2449)
        left, right = p[:half].sum(), p[1 + half:].sum()
        asymm = (left - right) / (left + right)
        if verbose:
            print("Chunk %s-%s success" % (idx_f, idx_t), flush=True)
    except Exception as exc:
        print(exc, flush=True)
        asymm = np.nan
    return (asymm, idx_f, idx_t)
