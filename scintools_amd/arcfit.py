"""Arc normalisation of the secondary spectrum on the GPU: the numerics of

* ``Dynspec.scale_dyn(scale='lambda')``  dynspec.py:3928-3959  -> ``scint_spline_resample``
* ``Dynspec.norm_sspec``                 dynspec.py:1920-2183  -> ``scint_norm_sspec`` +
                                                                   ``scint_masked_colavg`` (+ ``scint_row_nanmean``)
* ``Dynspec.fit_arc``                    dynspec.py:970-1313   -> ``scint_block_std`` + norm_sspec

The O(rows x columns) work (cubic-spline resample of every time column, the per-delay-row
``np.interp`` resample, masked means) runs in HIP kernels; the 1-D profile logic of ``fit_arc``
(Savitzky-Golay smoothing, peak walk, parabola fit) stays on the host exactly as in the
reference.  Plotting, ``velocity`` scaling, ``interp_nan`` (scipy griddata) and
``fit_spectrum`` (lmfit) are outside the accelerated path and raise ``NotImplementedError``.
"""
import ctypes

import numpy as np
import scipy.constants as sc
import torch
from scipy.signal import savgol_filter

from . import _lib
from .device import empty, ptr, require_gpu, stream_ptr, to_device, workspace


def is_valid(array):
    """scint_utils.py:87-91."""
    return np.isfinite(array) * (~np.isnan(array))


# ----------------------------------------------------------------------------
# scale_dyn(scale='lambda')
# ----------------------------------------------------------------------------
def _spline_system(x):
    """Thomas factors of the not-a-knot cubic spline in second-derivative form on knots x
    (ascending, len >= 4): interior rows i = 1..n-2,
        a_i M[i-1] + b_i M[i] + c_i M[i+1] = 6 ((y[i+1]-y[i])/h[i] - (y[i]-y[i-1])/h[i-1]),
    with M[0] and M[n-1] eliminated through the not-a-knot conditions."""
    n = len(x)
    h = np.diff(x)
    a = np.zeros(n)
    b = np.zeros(n)
    c = np.zeros(n)
    a[1:n - 1] = h[:n - 2]
    b[1:n - 1] = 2 * (h[:n - 2] + h[1:n - 1])
    c[1:n - 1] = h[1:n - 1]
    b[1] = (h[0] + h[1]) * (h[0] + 2 * h[1]) / h[1]
    c[1] = (h[1]**2 - h[0]**2) / h[1]
    a[1] = 0.0
    b[n - 2] = (h[n - 2] + h[n - 3]) * (h[n - 2] + 2 * h[n - 3]) / h[n - 3]
    a[n - 2] = (h[n - 3]**2 - h[n - 2]**2) / h[n - 3]
    c[n - 2] = 0.0
    inv = np.zeros(n)
    sup = np.zeros(n)
    inv[1] = 1.0 / b[1]
    sup[1] = c[1] * inv[1]
    for i in range(2, n - 1):
        den = b[i] - a[i] * sup[i - 1]
        inv[i] = 1.0 / den
        sup[i] = c[i] * inv[i]
    end = np.array([(h[0] + h[1]) / h[1], -h[0] / h[1],
                    (h[n - 2] + h[n - 3]) / h[n - 3], -h[n - 2] / h[n - 3]])
    return h, a, inv, sup, end


_SPLINE_BLOCK_ROWS = 128


def _spline_blocks(sub, inv, sup, n):
    """Frequency blocking of the two Thomas sweeps.  An error e in the forward value d_{i-1}
    reaches d_i as -sub[i]*inv[i]*e, one in M[i+1] reaches M[i] as -sup[i]*e: both factors are
    ~0.27 on a uniform axis, so a block may start `warm` rows early from zero.  `warm` is the
    shortest run over which EVERY window of factors multiplies to < 1e-22 (far below the 1e-16
    rounding of the values themselves); if the axis is so irregular that this needs more than a
    few blocks' worth of rows, fall back to one block (the plain sequential sweep)."""
    rows = n - 2
    if rows < 4 * _SPLINE_BLOCK_ROWS:
        return 0, 0
    tiny = 1e-300                                              # an exact zero factor (uniform ends) decouples
    lf = np.log(np.maximum(np.abs(sub[2:n - 1] * inv[2:n - 1]), tiny))   # forward factors, rows 2..n-2
    lb = np.log(np.maximum(np.abs(sup[1:n - 2]), tiny))                   # backward factors, rows 1..n-3
    target = np.log(1e-22)
    warm = 0
    for lg in (lf, lb):
        c = np.concatenate([[0.0], np.cumsum(lg)])
        w = 8
        while w < len(lg) and np.max(c[w:] - c[:-w]) > target:
            w += 8
        warm = max(warm, w)
    if warm > 2 * _SPLINE_BLOCK_ROWS:
        return 0, 0
    return _SPLINE_BLOCK_ROWS, int(warm)


def spline_resample_device(dyn_t, freqs, feq):
    """Cubic-spline (scipy ``interp1d(kind='cubic')``) resample of every time column of the
    device array dyn_t[nf, nt] from `freqs` to `feq`, rows flipped (dynspec.py:3948-3957)."""
    lib = _lib.load()
    require_gpu()
    freqs = np.asarray(freqs, dtype=float)
    nf, nt = (int(v) for v in dyn_t.shape)
    if nf != len(freqs):
        raise ValueError("x and y arrays must be equal in length along interpolation axis.")
    if nf < 4:
        raise ValueError("The number of derivatives at boundaries does not match: expected 1, got 0+0")
    d = np.diff(freqs)
    reverse = 0
    if np.all(d > 0):
        x = freqs
    elif np.all(d < 0):
        x, reverse = freqs[::-1].copy(), 1
    else:                                   # interp1d sorts an unsorted axis
        order = np.argsort(freqs, kind="mergesort")
        x = freqs[order]
        dyn_t = dyn_t[to_device(order.astype(np.int64), torch.int64)].contiguous()
        if np.any(np.diff(x) <= 0):
            raise ValueError("Expect x to not have duplicates")
    feq = np.asarray(feq, dtype=float)
    if feq.min() < x[0] or feq.max() > x[-1]:
        raise ValueError("A value in x_new is outside the interpolation range.")
    h, sub, inv, sup, end = _spline_system(x)
    idx = np.clip(np.searchsorted(x, feq, side="right") - 1, 0, nf - 2)
    hk = h[idx]
    A = (x[idx + 1] - feq) / hk
    B = (feq - x[idx]) / hk
    coef = np.stack([A, B, (A**3 - A) * hk**2 / 6.0, (B**3 - B) * hk**2 / 6.0], axis=1)
    block_rows, warm = _spline_blocks(sub, inv, sup, nf)
    if block_rows:
        # A NaN/inf pixel must poison its whole time column, as scipy's sequential solve (and
        # the unblocked sweep) does; warm-started blocks would confine it to one block.  One
        # device reduction tells: the mean is non-finite iff some pixel is.
        m = ctypes.c_double()
        _lib.check(lib.scint_mean(ptr(dyn_t), nf * nt, ctypes.byref(m), stream_ptr()), "scint_mean")
        if not np.isfinite(m.value):
            block_rows, warm = 0, 0
    ws = workspace.get(2 * 8 * nf * nt)
    out = empty((len(feq), nt), torch.float64)
    dev = lambda v: to_device(np.ascontiguousarray(v, dtype=float), torch.float64)
    h_t, sub_t, inv_t, sup_t, coef_t = dev(h), dev(sub), dev(inv), dev(sup), dev(coef)
    idx_t = to_device(idx.astype(np.int32), torch.int32)
    rc = lib.scint_spline_resample(ptr(dyn_t), nf, nt, reverse, ptr(h_t), ptr(sub_t), ptr(inv_t), ptr(sup_t),
                                   end.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), block_rows, warm,
                                   ptr(idx_t), ptr(coef_t), len(feq), ptr(out), ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_spline_resample")
    return out


def scale_dyn_lambda(self, spacing="auto"):
    """dyn(freq, t) -> dyn(lambda, t) in equal wavelength steps (dynspec.py:3928-3959).
    Sets ``lamdyn``, ``lam``, ``nlam``, ``dlam``."""
    freqs = np.array(self.freqs, dtype=float)
    lams = np.divide(sc.c, freqs * 10**6)
    step = np.abs(np.diff(lams))
    if spacing == "max":
        dlam = np.max(step)
    elif spacing == "median":
        dlam = np.median(step)
    elif spacing == "mean":
        dlam = np.mean(step)
    elif spacing == "min":
        dlam = np.min(step)
    elif spacing == "auto":
        dlam = (np.max(lams) - np.min(lams)) / len(freqs)
    else:
        raise UnboundLocalError("local variable 'dlam' referenced before assignment")
    lam_eq = np.arange(np.min(lams) + 1e-10, np.max(lams) - 1e-10, dlam)
    self.dlam = dlam
    feq = np.round(np.divide(sc.c, lam_eq) / 10**6, 6)
    if max(feq) > max(freqs):          # keep the rounded targets inside the band
        feq[np.argmax(feq)] = max(freqs)
    if min(feq) < min(freqs):
        feq[np.argmin(feq)] = min(freqs)
    dyn_t = to_device(np.asarray(self.dyn, dtype=float), torch.float64)
    lam_t = spline_resample_device(dyn_t, freqs, feq)
    type(self).lamdyn.park(self, lam_t)                 # stays in HBM until the host reads it
    self.lam = np.flipud(lam_eq)
    self.nlam = len(self.lam)


# ----------------------------------------------------------------------------
# norm_sspec
# ----------------------------------------------------------------------------
def _sspec_for(self, lamsteps):
    """Device tensor of the dB spectrum norm_sspec / fit_arc work on and its delay axis,
    computing the spectrum if absent (dynspec.py:1993-2023, 1069-1090)."""
    cls = type(self)
    if lamsteps:
        if not cls.lamsspec.present(self):
            self.calc_sspec(lamsteps=True)
        return cls.lamsspec.tensor(self), self.beta
    if not cls.sspec.present(self):
        self.calc_sspec()
    return cls.sspec.tensor(self), self.tdel


def norm_sspec(self, eta=None, delmax=None, plot=False, startbin=1, maxnormfac=5, minnormfac=0, cutmid=0,
               lamsteps=True, scrunched=True, plot_fit=True, ref_freq=1400, velocity=False, numsteps=None,
               filename=None, display=True, weighted=True, unscrunched=True, logsteps=False, powerspec=True,
               interp_nan=False, fit_spectrum=False, powerspec_cut=False, figsize=(9, 9),
               subtract_artefacts=False, dpi=200):
    """Normalise the Doppler axis by the arc curvature and scrunch in delay
    (dynspec.py:1920-2183).  Sets ``normsspecavg``, ``normsspec`` (2-D masked array, copied
    from the device on first access), ``normsspec_tdel``, ``normsspec_fdop``, ``powerspectrum``,
    ``weights``, ``mask``."""
    if plot:
        raise NotImplementedError("plotting is outside the accelerated hot path")
    if velocity:
        raise NotImplementedError("velocity scaling needs Dynspec.scale_dyn('velocity') (outside the hot path)")
    if interp_nan:
        raise NotImplementedError("interp_nan (scipy.interpolate.griddata) is outside the accelerated path")
    if fit_spectrum:
        raise NotImplementedError("fit_spectrum needs lmfit and is outside the accelerated path")
    lib = _lib.load()
    require_gpu()
    if not hasattr(self, "tdel"):
        self.calc_sspec(lamsteps=lamsteps)
    delmax = np.max(self.tdel) if delmax is None else delmax
    held = getattr(self, "_arc_dev_cache", None)          # a running fit_arc: (tensor, yaxis, lamsteps)
    if held is not None and held[2] == bool(lamsteps):
        sspec_t, yaxis = held[0], held[1]
    else:
        sspec_t, yaxis = _sspec_for(self, lamsteps)
    if eta is None:
        if not hasattr(self, "betaeta" if lamsteps else "eta"):
            self.fit_arc(lamsteps=lamsteps, delmax=delmax, plot=plot, startbin=startbin)
        eta = self.betaeta if lamsteps else self.eta
    elif not lamsteps:                                  # dynspec.py:2033-2038
        c = 299792458.0
        beta_to_eta = c * 1e6 / ((ref_freq * 10**6)**2)
        eta = eta / (self.freq / ref_freq)**2
        eta = eta * beta_to_eta
    eta = float(eta)
    fdop = np.asarray(self.fdop, dtype=float)
    if np.any(np.diff(fdop) <= 0):
        raise ValueError("norm_sspec: the fdop axis must be ascending")
    ind = int(np.argmin(abs(self.tdel - delmax)))
    nrow_all, nc = (int(v) for v in sspec_t.shape)
    row0 = int(startbin)
    nr = len(range(nrow_all)[startbin:ind])
    tdel = np.array(yaxis[startbin:ind], dtype=float)
    if nr < 1:
        raise IndexError("index -1 is out of bounds for axis 0 with size 0")
    cut_lo = int(nc / 2 - np.floor(cutmid / 2))
    cut_hi = int(nc / 2 + np.floor(cutmid / 2))
    fdop_t = to_device(fdop, torch.float64)
    yaxis_t = to_device(np.asarray(yaxis, dtype=float), torch.float64)
    offset_t = None
    if subtract_artefacts:                              # dynspec.py:2057-2063
        colsel = (np.abs(fdop) > 0.9 * np.max(fdop)).astype(np.uint8)
        colsel_t = to_device(colsel, torch.uint8)
        resp_t = empty((nr,), torch.float64)
        _lib.check(lib.scint_row_nanmean(ptr(sspec_t), nc, nc, row0, nr, ptr(colsel_t), cut_lo, cut_hi,
                                         ptr(resp_t), stream_ptr()), "scint_row_nanmean")
        resp = resp_t.cpu().numpy()
        resp -= np.median(resp)
        offset_t = to_device(resp, torch.float64)
    maxfdop = maxnormfac * np.sqrt(tdel[-1] / eta)
    if maxfdop > max(fdop):
        maxfdop = max(fdop)
    nfdop = 2 * len(fdop[abs(fdop) <= maxfdop]) if numsteps is None else numsteps
    if nfdop % 2 != 0:
        nfdop += 1
    fdoplin = None
    if logsteps:                                        # dynspec.py:2076-2083
        fdoplin = np.abs(np.linspace(-maxnormfac, maxnormfac, int(nfdop)))
        fdop_pos = 10**np.linspace(np.log10(np.min(fdoplin)), np.log10(np.max(fdoplin)), int(nfdop / 2))
        fdopnew = np.concatenate((-np.flip(fdop_pos, axis=0), fdop_pos))
    else:
        fdopnew = np.linspace(-maxnormfac, maxnormfac, int(nfdop))
    if minnormfac > 0:
        fdopnew = fdopnew[np.argwhere(np.abs(fdopnew) > minnormfac)]
        if logsteps:
            raise ValueError("Mask and data not compatible: logsteps with minnormfac > 0 "
                             "(the reference fails the same way, dynspec.py:2117)")
    # the first (smallest-delay) row selects the fewest Doppler bins; np.interp raises on none
    if not np.any(abs(fdop) <= maxnormfac * np.sqrt(np.min(tdel) / eta)):
        raise ValueError("array of sample points is empty")
    x = np.ascontiguousarray(np.ravel(fdopnew), dtype=float)
    nx = len(x)
    x_t = to_device(x, torch.float64)
    xlin_t = to_device(np.ascontiguousarray(fdoplin, dtype=float), torch.float64) if logsteps else None
    norm_t = empty((nr, nx), torch.float64)
    mask_t = empty((nr, nx), torch.uint8)
    pow_t = empty((nr,), torch.float64)
    rc = lib.scint_norm_sspec(ptr(sspec_t), nc, nc, ptr(fdop_t), ptr(yaxis_t), row0, nr, eta,
                              float(maxnormfac), cut_lo, cut_hi, ptr(offset_t), ptr(x_t), ptr(xlin_t), nx,
                              ptr(norm_t), ptr(mask_t), ptr(pow_t), stream_ptr())
    _lib.check(rc, "scint_norm_sspec")
    self.powerspectrum = np.ma.masked_invalid(pow_t.cpu().numpy())
    xdata = np.sqrt(tdel)
    ydata = np.sqrt(tdel) * self.powerspectrum
    xdata = xdata[~np.isnan(xdata)]
    ydata = ydata[~np.isnan(ydata)]
    alpha = -11 / 3                                     # dynspec.py:2133-2137
    index = np.argmin(np.abs(xdata - 10))
    amp = ydata[index] * xdata[index]**-alpha
    wn = np.min(ydata)
    arc_spectrum = amp * xdata**alpha
    if weighted:
        self.weights = 10 * np.log10(arc_spectrum)
    else:
        self.weights = np.ones(np.shape(arc_spectrum))
    wts = np.ma.filled(np.ma.array(self.weights, dtype=float), np.nan).squeeze()
    if wts.shape != (nr,):
        raise ValueError("Length of weights not compatible with specified axis.")
    rowsel_t = None
    if powerspec_cut:                                   # dynspec.py:2171-2178
        rowsel = np.zeros(nr, dtype=np.uint8)
        rowsel[np.argwhere(np.ma.filled(arc_spectrum > wn, False)).ravel()] = 1
        rowsel_t = to_device(rowsel, torch.uint8)
    w_t = to_device(np.ascontiguousarray(wts), torch.float64)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_masked_colavg_workspace_bytes(nr, nx, ctypes.byref(need)), "masked_colavg_workspace_bytes")
    ws = workspace.get(need.value)
    avg_t = empty((nx,), torch.float64)
    none_t = empty((nx,), torch.uint8)
    rc = lib.scint_masked_colavg(ptr(norm_t), ptr(mask_t), nr, nx, ptr(w_t), ptr(rowsel_t), ptr(avg_t),
                                 ptr(none_t), ptr(ws), ws.numel(), stream_ptr())
    _lib.check(rc, "scint_masked_colavg")
    # a column with no unmasked entry is masked, with numpy.ma's 0.0 left under the mask
    self.normsspecavg = np.ma.array(avg_t.cpu().numpy(), mask=none_t.cpu().numpy().astype(bool))
    self._normsspec_dev = (norm_t, mask_t)
    self._normsspec_host = None
    self._mask_host = None
    self.normsspec_tdel = tdel
    self.normsspec_fdop = fdopnew
    return


def normsspec_host(self):
    """Materialise the 2-D normalised spectrum (masked array) from the device on first use."""
    if getattr(self, "_normsspec_host", None) is None:
        dev = getattr(self, "_normsspec_dev", None)
        if dev is None:
            raise AttributeError("'Dynspec' object has no attribute 'normsspec'")
        self._normsspec_host = np.ma.array(dev[0].cpu().numpy(), mask=dev[1].cpu().numpy().astype(bool))
    return self._normsspec_host


# ----------------------------------------------------------------------------
# parabola fits (scint_models.py:300-347)
# ----------------------------------------------------------------------------
def fit_parabola(x, y):
    """Peak of the least-squares parabola and its error (scint_models.py:300-326)."""
    ptp = np.ptp(x)
    x = x * (1000 / ptp)
    params, pcov = np.polyfit(x, y, 2, cov=True)
    yfit = params[0] * np.power(x, 2) + params[1] * x + params[2]
    errors = [np.absolute(pcov[i][i])**0.5 for i in range(len(params))]
    peak = -params[1] / (2 * params[0])
    peak_error = np.sqrt((errors[1]**2) * ((1 / (2 * params[0]))**2) + (errors[0]**2) * ((params[1] / 2)**2))
    return yfit, peak * (ptp / 1000), peak_error * (ptp / 1000)


def fit_log_parabola(x, y):
    """The same in log(x) (scint_models.py:329-347)."""
    logx = np.log(x)
    ptp = np.ptp(logx)
    x = logx * (1000 / ptp)
    yfit, peak, peak_error = fit_parabola(x, y)
    frac_error = peak_error / peak
    peak = np.e**(peak * ptp / 1000)
    return yfit, peak, frac_error * peak


# ----------------------------------------------------------------------------
# fit_arc
# ----------------------------------------------------------------------------
def fit_arc(self, asymm=False, plot=False, delmax=None, numsteps=1e4, startbin=3, cutmid=3, lamsteps=False,
            etamax=None, etamin=None, low_power_diff=-1, high_power_diff=-0.5, ref_freq=1400,
            constraint=[0, np.inf], nsmooth=5, efac=1, filename=None, noise_error=True, display=True,
            figN=None, log_parabola=False, logsteps=False, plot_spec=False, fit_spectrum=False,
            subtract_artefacts=False, figsize=(9, 9), dpi=200, velocity=False, weighted=False):
    """Find the arc curvature with maximum power along it (dynspec.py:970-1313).  Sets
    ``eta / etaerr / etaerr2`` (or ``betaeta...`` with lamsteps, ``..._left / _right`` with
    asymm), ``noise``, ``eta_array``, ``norm_sspec_avg*``, ``prob_eta_peak*``, ``norm_delmax``."""
    if plot or plot_spec:
        raise NotImplementedError("plotting is outside the accelerated hot path")
    if velocity:
        raise NotImplementedError("velocity scaling needs Dynspec.scale_dyn('velocity') (outside the hot path)")
    lib = _lib.load()
    require_gpu()
    if not hasattr(self, "tdel"):
        self.calc_sspec()
    delmax = np.max(self.tdel) if delmax is None else delmax
    sspec_t, yaxis_full = _sspec_for(self, lamsteps)
    yaxis = np.array(yaxis_full, dtype=float)
    ind = int(np.argmin(abs(self.tdel - delmax)))
    ymax = self.beta[ind]                                # dynspec.py:1092 (needs a lamsteps spectrum)
    nr, nc = (int(v) for v in sspec_t.shape)
    # noise of the spectrum: std of the outer half in delay, centre columns excluded (dynspec.py:1097-1101)
    c_hi = int(nc / 2 + np.ceil(cutmid / 2))
    c_lo = int(nc / 2 - np.floor(cutmid / 2))
    std_t = empty((1,), torch.float64)
    ws = workspace.get(8 * 1032)
    rc = lib.scint_block_std(ptr(sspec_t), nc, nc, int(nr / 2), nr, c_lo, c_hi, ptr(std_t), ptr(ws),
                             ws.numel(), stream_ptr())
    _lib.check(rc, "scint_block_std")
    noise = float(std_t.cpu().numpy()[0])
    yaxis = yaxis[0:ind]
    noise = np.sqrt(np.sum(np.power(noise, 2))) / np.sqrt(len(yaxis) * 2)
    self.noise = noise
    if etamax is None:
        etamax = ymax / ((self.fdop[1] - self.fdop[0]) * cutmid)**2
    if etamin is None:
        etamin = (yaxis[1] - yaxis[0]) * startbin / (max(self.fdop))**2
    try:
        len(etamin)
        etamin_array = np.array(etamin).squeeze()
        etamax_array = np.array(etamax).squeeze()
    except TypeError:
        etamin_array = np.array([etamin])
        etamax_array = np.array([etamax])
    sqrt_eta_all = np.linspace(np.sqrt(np.min(etamin_array)), np.sqrt(np.max(etamax_array)), int(numsteps))
    self._arc_dev_cache = (sspec_t, yaxis_full, bool(lamsteps))
    try:
        for iarc in range(len(etamin_array)):
            if len(etamin_array) != 1:
                etamin = etamin_array.squeeze()[iarc]
                etamax = etamax_array.squeeze()[iarc]
            if not lamsteps:                             # dynspec.py:1140-1148
                c = 299792458.0
                beta_to_eta = c * 1e6 / ((ref_freq * 10**6)**2)
                etamax = etamax / (self.freq / ref_freq)**2
                etamax = etamax * beta_to_eta
                etamin = etamin / (self.freq / ref_freq)**2
                etamin = etamin * beta_to_eta
                constraint = constraint / (self.freq / ref_freq)**2
                constraint = constraint * beta_to_eta
            sqrt_eta = sqrt_eta_all[(sqrt_eta_all <= np.sqrt(etamax)) * (sqrt_eta_all >= np.sqrt(etamin))]
            # normalised spectrum with etamin as the normalisation: 1/fdop_norm**2 scans eta
            self.norm_sspec(eta=etamin, delmax=delmax, plot=False, startbin=startbin, maxnormfac=1,
                            cutmid=cutmid, lamsteps=lamsteps, scrunched=True, logsteps=logsteps,
                            plot_fit=False, numsteps=len(sqrt_eta), fit_spectrum=fit_spectrum,
                            subtract_artefacts=subtract_artefacts, velocity=velocity, weighted=weighted)
            norm_sspec_avg_all = self.normsspecavg.squeeze()
            etafrac_array = self.normsspec_fdop
            ind1 = np.argwhere(etafrac_array >= 0)
            ind2 = np.argwhere(etafrac_array < 0)
            if asymm:
                profiles = [np.array(norm_sspec_avg_all[ind1]), np.flip(norm_sspec_avg_all[ind2], axis=0)]
            else:
                profiles = [np.add(norm_sspec_avg_all[ind1], np.flip(norm_sspec_avg_all[ind2], axis=0)) / 2]
            etafrac_array_avg_orig = 1 / etafrac_array[ind1].squeeze()
            for dummy, spec in enumerate(profiles):
                # np.array() drops the mask and keeps what numpy.ma left under it, as the reference does
                spec = np.array(spec).squeeze()
                filt_ind = is_valid(spec)
                spec = np.flip(spec[filt_ind], axis=0)
                etafrac_array_avg = np.flip(etafrac_array_avg_orig[filt_ind], axis=0)
                etaArray = etamin * etafrac_array_avg**2
                keep = np.argwhere(etaArray < etamax)
                etaArray = etaArray[keep].squeeze()
                spec = spec[keep].squeeze()
                smooth = savgol_filter(spec, nsmooth, 1)
                indrange = np.argwhere((etaArray > constraint[0]) * (etaArray < constraint[1]))
                pk = np.argmin(np.abs(smooth - np.max(smooth[indrange])))
                max_power = smooth[pk]
                power, i1 = max_power, 1                 # dynspec.py:1222-1233
                while power > max_power + low_power_diff and pk + i1 < len(smooth) - 1:
                    i1 += 1
                    power = smooth[pk - i1]
                power, i2 = max_power, 1
                while power > max_power + high_power_diff and pk + i2 < len(smooth) - 1:
                    i2 += 1
                    power = smooth[pk + i2]
                xdata = etaArray[int(pk - i1):int(pk + i2)]
                ydata = spec[int(pk - i1):int(pk + i2)]
                if log_parabola:
                    yfit, eta, etaerr = fit_log_parabola(xdata, ydata)
                else:
                    yfit, eta, etaerr = fit_parabola(xdata, ydata)
                if np.mean(np.gradient(np.diff(yfit))) > 0:
                    raise ValueError('Fit returned a forward parabola.')
                etaerr2 = etaerr                         # error from the parabola fit
                if noise_error:                          # dynspec.py:1250-1265
                    power, i1 = max_power, 1
                    while power > (max_power - noise) and (pk - i1 > 1):
                        power = smooth[pk - i1]
                        i1 += 1
                    power, i2 = max_power, 1
                    while power > (max_power - noise) and (pk + i2 < len(smooth) - 1):
                        i2 += 1
                        power = smooth[pk + i2]
                    etaerr = np.abs(etaArray[int(pk - i1)] - etaArray[int(pk + i2)]) / 2
                self.eta_array = etaArray
                sigma = self.noise * efac
                prob = 1 / (sigma * np.sqrt(2 * np.pi)) * np.exp(-0.5 * ((spec - np.max(spec)) / sigma)**2)
                if asymm:
                    if dummy == 0:
                        self.norm_sspec_avg1, self.prob_eta_peak1 = spec, prob
                    else:
                        self.norm_sspec_avg2, self.prob_eta_peak2 = spec, prob
                else:
                    self.norm_sspec_avg, self.prob_eta_peak = spec, prob
                if iarc == 0:                            # save primary (dynspec.py:1283-1313)
                    stem = "betaeta" if lamsteps else "eta"
                    side = ("_left" if dummy == 0 else "_right") if asymm else ""
                    setattr(self, stem + side, eta)
                    setattr(self, stem + "err" + side, etaerr / np.sqrt(2))
                    setattr(self, stem + "err2" + side, etaerr2 / np.sqrt(2))
            self.norm_delmax = delmax
    finally:
        self._arc_dev_cache = None
