"""CPU oracle for the arc-normalisation row -- TEST INFRASTRUCTURE.

Plain NumPy/SciPy restatement of

* ``Dynspec.scale_dyn(scale='lambda')``   dynspec.py:3928-3959  (equal-wavelength resample)
* the ``lamsteps`` axis of ``calc_sspec``  dynspec.py:3643-3650, 3703-3704
* ``Dynspec.norm_sspec``                   dynspec.py:1993-2183  (numerics only, no plotting)
* ``Dynspec.fit_arc``                      dynspec.py:1066-1313
* ``scint_models.fit_parabola / fit_log_parabola``  scint_models.py:300-347

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  ``interp_nan``, ``fit_spectrum`` (lmfit) and ``velocity`` are not restated.

PARITY PIN: checked against ``tests/golden/arcfit.npz`` -- outputs of the unmodified
reference methods run in the build container on a seeded reference ``Simulation``
(tests/golden/make_golden.py, ``gen_arcfit``).
"""
import numpy as np
import scipy.constants as sc
from scipy.interpolate import interp1d
from scipy.signal import savgol_filter

from . import sspec_oracle


def is_valid(array):
    """scint_utils.py:87-91."""
    return np.isfinite(array) * (~np.isnan(array))


# ---------------------------------------------------------------------------
# scale_dyn(scale='lambda')
# ---------------------------------------------------------------------------
def scale_dyn_lambda(dyn, freqs, spacing="auto"):
    """Resample dyn[nf, nt] from equal frequency steps to equal wavelength steps with a
    cubic spline down every time column (dynspec.py:3928-3959).
    Returns (lamdyn, lam, dlam); lam is descending-frequency ordered (flipud)."""
    arin = np.array(dyn, dtype=float)
    nf, nt = arin.shape
    freqs = np.array(freqs, dtype=float)
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
        raise ValueError(spacing)
    lam_eq = np.arange(np.min(lams) + 1e-10, np.max(lams) - 1e-10, dlam)
    feq = np.round(np.divide(sc.c, lam_eq) / 10**6, 6)
    # keep the rounded targets inside the sampled band (dynspec.py:3952-3955)
    if max(feq) > max(freqs):
        feq[np.argmax(feq)] = max(freqs)
    if min(feq) < min(freqs):
        feq[np.argmin(feq)] = min(freqs)
    arout = np.zeros([len(lam_eq), int(nt)])
    for it in range(nt):
        arout[:, it] = interp1d(freqs, arin[:, it], kind="cubic")(feq)
    return np.flipud(arout), np.flipud(lam_eq), dlam


def calc_sspec_lam(dyn, freqs, dt, df, **kw):
    """calc_sspec(lamsteps=True): secondary spectrum of the wavelength-scaled dynspec
    (dynspec.py:3643-3650) with the conjugate-wavelength axis beta (dynspec.py:3703-3704).
    Returns dict(lamdyn, lam, dlam, fdop, tdel, beta, lamsspec)."""
    lamdyn, lam, dlam = scale_dyn_lambda(dyn, freqs)
    fdop, tdel, sec = sspec_oracle.calc_sspec(lamdyn, dt, df, **kw)
    nrfft, _ = sspec_oracle.fft_lengths(*lamdyn.shape)
    td = np.arange(0, len(tdel))
    beta = np.divide(td, (nrfft * dlam))
    return dict(lamdyn=lamdyn, lam=lam, dlam=dlam, fdop=fdop, tdel=tdel, beta=beta, lamsspec=sec)


# ---------------------------------------------------------------------------
# norm_sspec
# ---------------------------------------------------------------------------
def norm_sspec(sspec, yaxis, tdel_axis, fdop, freq, eta, delmax=None, startbin=1, maxnormfac=5,
               minnormfac=0, cutmid=0, lamsteps=True, ref_freq=1400, numsteps=None, weighted=True,
               logsteps=False, powerspec_cut=False, subtract_artefacts=False):
    """Normalise the Doppler axis of a secondary spectrum by an arc curvature and scrunch in
    delay (dynspec.py:1993-2183).  `sspec` is the dB spectrum matching `lamsteps`, `yaxis` its
    delay axis (beta or tdel), `tdel_axis` is always self.tdel (used for the delmax cut).
    Returns a dict with the attributes the reference sets."""
    delmax = np.max(tdel_axis) if delmax is None else delmax
    sspec = np.array(sspec, dtype=float)
    if not lamsteps:                                               # dynspec.py:2033-2038
        c = 299792458.0
        beta_to_eta = c * 1e6 / ((ref_freq * 10**6)**2)
        eta = eta / (freq / ref_freq)**2
        eta = eta * beta_to_eta
    ind = np.argmin(abs(tdel_axis - delmax))
    sspec = sspec[startbin:ind, :]
    nr, nc = np.shape(sspec)
    sspec[:, int(nc / 2 - np.floor(cutmid / 2)):int(nc / 2 + np.floor(cutmid / 2))] = np.nan
    tdel = np.array(yaxis[startbin:ind])
    if subtract_artefacts:                                         # dynspec.py:2057-2063
        outer = np.argwhere(np.abs(fdop) > 0.9 * np.max(fdop))
        delay_response = np.nanmean(sspec[:, outer], axis=1)
        delay_response -= np.median(delay_response)
        sspec = np.subtract(sspec, delay_response)
    maxfdop = maxnormfac * np.sqrt(tdel[-1] / eta)
    if maxfdop > max(fdop):
        maxfdop = max(fdop)
    nfdop = 2 * len(fdop[abs(fdop) <= maxfdop]) if numsteps is None else numsteps
    if nfdop % 2 != 0:
        nfdop += 1
    if logsteps:                                                   # dynspec.py:2076-2083
        fdoplin = np.abs(np.linspace(-maxnormfac, maxnormfac, int(nfdop)))
        fdop_pos = 10**np.linspace(np.log10(np.min(fdoplin)), np.log10(np.max(fdoplin)), int(nfdop / 2))
        fdopnew = np.concatenate((-np.flip(fdop_pos, axis=0), fdop_pos))
    else:
        fdopnew = np.linspace(-maxnormfac, maxnormfac, int(nfdop))
    if minnormfac > 0:
        fdopnew = fdopnew[np.argwhere(np.abs(fdopnew) > minnormfac)]
    rows, rows_lin, mask = [], [], []
    for ii in range(len(tdel)):                                    # dynspec.py:2093-2107
        scale = np.sqrt(tdel[ii] / eta)
        sel = abs(fdop) <= maxnormfac * scale
        ifdop = fdop[sel] / scale
        isspec = sspec[ii, sel]
        if logsteps:
            rows_lin.append(np.interp(fdoplin, ifdop, isspec))
        rows.append(np.interp(fdopnew, ifdop, isspec))
        mask.append(np.abs(fdopnew) > np.max(np.abs(ifdop)))
    mask = np.array(mask).squeeze()
    norm = np.array(rows).squeeze()
    if logsteps:                                                   # dynspec.py:2115-2123
        lin = np.ma.array(np.array(rows_lin).squeeze(), mask=mask)   # shares `mask`
        mask += np.isnan(norm)
        norm = np.ma.array(norm, mask=mask)
        powerspectrum = np.ma.mean(np.power(10, lin / 10), axis=1)
    else:
        mask += np.isnan(norm)
        norm = np.ma.array(norm, mask=mask)
        powerspectrum = np.ma.mean(np.power(10, norm / 10), axis=1)
    xdata = np.sqrt(tdel)
    ydata = np.sqrt(tdel) * powerspectrum
    xdata = xdata[~np.isnan(xdata)]
    ydata = ydata[~np.isnan(ydata)]
    alpha = -11 / 3                                                # dynspec.py:2133-2137
    index = np.argmin(np.abs(xdata - 10))
    amp = ydata[index] * xdata[index]**-alpha
    wn = np.min(ydata)
    arc_spectrum = amp * xdata**alpha
    weights = 10 * np.log10(arc_spectrum) if weighted else np.ones(np.shape(arc_spectrum))
    if powerspec_cut:                                              # dynspec.py:2171-2178
        keep = np.argwhere(arc_spectrum > wn)
        avg = np.ma.average(norm[keep, :], axis=0, weights=weights[keep].squeeze()).squeeze()
    else:
        avg = np.ma.average(norm, axis=0, weights=weights.squeeze()).squeeze()
    return dict(normsspecavg=avg, normsspec=norm, normsspec_tdel=tdel, normsspec_fdop=fdopnew,
                powerspectrum=powerspectrum, weights=weights, mask=mask, eta_used=eta)


# ---------------------------------------------------------------------------
# parabola fits (scint_models.py:300-347)
# ---------------------------------------------------------------------------
def fit_parabola(x, y):
    ptp = np.ptp(x)
    x = x * (1000 / ptp)
    params, pcov = np.polyfit(x, y, 2, cov=True)
    yfit = params[0] * np.power(x, 2) + params[1] * x + params[2]
    errors = [np.absolute(pcov[i][i])**0.5 for i in range(len(params))]
    peak = -params[1] / (2 * params[0])
    peak_error = np.sqrt((errors[1]**2) * ((1 / (2 * params[0]))**2) +
                         (errors[0]**2) * ((params[1] / 2)**2))
    return yfit, peak * (ptp / 1000), peak_error * (ptp / 1000)


def fit_log_parabola(x, y):
    logx = np.log(x)
    ptp = np.ptp(logx)
    x = logx * (1000 / ptp)
    yfit, peak, peak_error = fit_parabola(x, y)
    frac_error = peak_error / peak
    peak = np.e**(peak * ptp / 1000)
    return yfit, peak, frac_error * peak


# ---------------------------------------------------------------------------
# fit_arc
# ---------------------------------------------------------------------------
def fit_arc(sspec, yaxis, tdel_axis, beta_axis, fdop, freq, asymm=False, delmax=None, numsteps=1e4,
            startbin=3, cutmid=3, lamsteps=False, etamax=None, etamin=None, low_power_diff=-1,
            high_power_diff=-0.5, ref_freq=1400, constraint=(0, np.inf), nsmooth=5, efac=1,
            noise_error=True, log_parabola=False, logsteps=False, subtract_artefacts=False,
            weighted=False):
    """Curvature of the arc with the most power along it (dynspec.py:1066-1313), first arc only
    (scalar etamin / etamax).  `sspec`/`yaxis` match `lamsteps`; `tdel_axis`/`beta_axis` are
    self.tdel and self.beta.  Returns a dict of the attributes the reference sets."""
    delmax = np.max(tdel_axis) if delmax is None else delmax
    sspec = np.array(sspec, dtype=float)
    yaxis = np.array(yaxis, dtype=float)
    ind = np.argmin(abs(tdel_axis - delmax))
    ymax = beta_axis[ind]
    nr, nc = np.shape(sspec)
    a = np.array(sspec[int(nr / 2):, int(nc / 2 + np.ceil(cutmid / 2)):].ravel())   # dynspec.py:1097-1101
    b = np.array(sspec[int(nr / 2):, 0:int(nc / 2 - np.floor(cutmid / 2))].ravel())
    noise = np.std(np.concatenate((a, b)))
    yaxis = yaxis[0:ind]
    noise = np.sqrt(np.sum(np.power(noise, 2))) / np.sqrt(len(yaxis) * 2)
    if etamax is None:
        etamax = ymax / ((fdop[1] - fdop[0]) * cutmid)**2
    if etamin is None:
        etamin = (yaxis[1] - yaxis[0]) * startbin / (max(fdop))**2
    sqrt_eta_all = np.linspace(np.sqrt(etamin), np.sqrt(etamax), int(numsteps))
    constraint = np.asarray(constraint, dtype=float)
    if not lamsteps:                                               # dynspec.py:1140-1148
        c = 299792458.0
        beta_to_eta = c * 1e6 / ((ref_freq * 10**6)**2)
        etamax = etamax / (freq / ref_freq)**2 * beta_to_eta
        etamin = etamin / (freq / ref_freq)**2 * beta_to_eta
        constraint = constraint / (freq / ref_freq)**2 * beta_to_eta
    sqrt_eta = sqrt_eta_all[(sqrt_eta_all <= np.sqrt(etamax)) * (sqrt_eta_all >= np.sqrt(etamin))]
    ns = norm_sspec(sspec, np.array(beta_axis if lamsteps else tdel_axis), tdel_axis, fdop, freq,
                    eta=etamin, delmax=delmax, startbin=startbin, maxnormfac=1, cutmid=cutmid,
                    lamsteps=lamsteps, ref_freq=ref_freq, numsteps=len(sqrt_eta), logsteps=logsteps,
                    subtract_artefacts=subtract_artefacts, weighted=weighted)
    prof = ns["normsspecavg"].squeeze()
    frac = ns["normsspec_fdop"]
    pos = np.argwhere(frac >= 0)
    neg = np.argwhere(frac < 0)
    if asymm:
        sides = [np.array(prof[pos]), np.flip(prof[neg], axis=0)]
    else:
        sides = [np.add(prof[pos], np.flip(prof[neg], axis=0)) / 2]
    inv_frac = 1 / frac[pos].squeeze()
    out = dict(noise=noise, norm=ns, sides=[])
    for spec in sides:
        spec = np.array(spec).squeeze()
        ok = is_valid(spec)
        spec = np.flip(spec[ok], axis=0)
        fr = np.flip(inv_frac[ok], axis=0)
        eta_array = etamin * fr**2
        cut = np.argwhere(eta_array < etamax)
        eta_array = eta_array[cut].squeeze()
        spec = spec[cut].squeeze()
        smooth = savgol_filter(spec, nsmooth, 1)
        inrange = np.argwhere((eta_array > constraint[0]) * (eta_array < constraint[1]))
        ipk = np.argmin(np.abs(smooth - np.max(smooth[inrange])))
        max_power = smooth[ipk]
        power, i1 = max_power, 1                                   # dynspec.py:1222-1233
        while power > max_power + low_power_diff and ipk + i1 < len(smooth) - 1:
            i1 += 1
            power = smooth[ipk - i1]
        power, i2 = max_power, 1
        while power > max_power + high_power_diff and ipk + i2 < len(smooth) - 1:
            i2 += 1
            power = smooth[ipk + i2]
        xdata = eta_array[int(ipk - i1):int(ipk + i2)]
        ydata = spec[int(ipk - i1):int(ipk + i2)]
        yfit, eta, etaerr = (fit_log_parabola if log_parabola else fit_parabola)(xdata, ydata)
        if np.mean(np.gradient(np.diff(yfit))) > 0:
            raise ValueError("Fit returned a forward parabola.")
        etaerr2 = etaerr
        if noise_error:                                            # dynspec.py:1250-1265
            power, i1 = max_power, 1
            while power > (max_power - noise) and (ipk - i1 > 1):
                power = smooth[ipk - i1]
                i1 += 1
            power, i2 = max_power, 1
            while power > (max_power - noise) and (ipk + i2 < len(smooth) - 1):
                i2 += 1
                power = smooth[ipk + i2]
            etaerr = np.abs(eta_array[int(ipk - i1)] - eta_array[int(ipk + i2)]) / 2
        sigma = noise * efac
        prob = 1 / (sigma * np.sqrt(2 * np.pi)) * np.exp(-0.5 * ((spec - np.max(spec)) / sigma)**2)
        out["sides"].append(dict(eta=eta, etaerr=etaerr / np.sqrt(2), etaerr2=etaerr2 / np.sqrt(2),
                                 eta_array=eta_array, spec=spec, smooth=smooth, prob=prob,
                                 xdata=xdata, ydata=ydata, yfit=yfit))
    return out
