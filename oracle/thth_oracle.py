"""CPU oracle for the theta-theta half of the hot path -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``scintools_amd``) never does and
fails loudly when its HIP library is missing.

It is a plain-float64 NumPy/SciPy restatement of the reference's algorithm
(``/root/reference/scintools/ththmod.py``), unit-free: every array is already
in the unit the reference coerces to with ``unit_checks`` --
``tau`` [us], ``fd`` [mHz], ``eta`` [s**3 == us/mHz**2], ``edges`` [mHz],
``freq`` [MHz], ``time`` [s].  The order of floating-point operations follows
the reference line by line because the nearest-bin gather is decided by a
floor (one ulp flips a pixel).

PARITY PIN: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which were produced by running the UNMODIFIED reference
modules in the build container (``tests/golden/make_golden.py``; astropy is
replaced by the small unit-tracking shim under ``tests/golden/refshim``).  The
reference ships no tests of its own; its one documented known answer
(eta ~ 44 s**3 on Sample_Data.npz, docs/source/tutorials/thth_intro.rst:101-103)
is checked there too.
"""
import numpy as np
from scipy.optimize import curve_fit
from scipy.sparse.linalg import eigsh


# --------------------------------------------------------------------------
# axes / helpers
# --------------------------------------------------------------------------
def chi_par(x, A, x0, C):
    """Parabola used for the eigenvalue-peak fit (ththmod.py:38-53)."""
    return A * (x - x0) ** 2 + C


def fft_axis(x, scale, pad=0):
    """Conjugate axis of ``x`` (ththmod.py:473-493).

    ``scale`` is the factor of ``.to_value(unit)``: 1000.0 for s -> mHz and
    1.0 for MHz -> us.  The multiply is skipped for 1.0, as astropy does.
    """
    fx = np.fft.fftfreq((pad + 1) * x.shape[0], x[1] - x[0])
    if scale != 1.0:
        fx = fx * scale
    return np.fft.fftshift(fx)


def theta_centres(edges):
    """Bin centres, shifted so the smallest |theta| is exactly 0 (ththmod.py:83-84)."""
    th = (edges[1:] + edges[:-1]) / 2
    th -= th[np.abs(th) == np.abs(th).min()]
    return th


def min_edges(fd_lim, fd, tau, eta, factor=2):
    """Smallest ``edges`` oversampling the CS by ``factor`` (ththmod.py:1671-1705)."""
    dtau_lim = (tau[1] - tau[0]) / factor
    dtau_lim /= 2 * eta * fd_lim
    dfd_lim = (fd[1] - fd[0]) / factor
    npoints = (2 * fd_lim) // (min(dfd_lim, dtau_lim))
    npoints += np.mod(npoints, 2)
    return np.linspace(-fd_lim, fd_lim, int(npoints))


# --------------------------------------------------------------------------
# CS -> theta-theta (gather)
# --------------------------------------------------------------------------
def thth_index_maps(tau, fd, eta, edges):
    """Integer gather maps of thth_map (ththmod.py:83-97, 103).

    Returns (tau_inv, fd_inv, pnts, th1, th2) with th1[i, j] = theta_j and
    th2[i, j] = theta_i.
    """
    th_cents = theta_centres(edges)
    th1 = np.ones((th_cents.shape[0], th_cents.shape[0])) * th_cents
    th2 = th1.T
    dtau = np.diff(tau).mean()
    dfd = np.diff(fd).mean()
    tau_inv = (((eta * (th1**2 - th2**2)) - tau[0] + dtau / 2) // dtau).astype(int)
    fd_inv = (((th1 - th2) - fd[0] + dfd / 2) // dfd).astype(int)
    pnts = (tau_inv > 0) * (tau_inv < tau.shape[0]) * (fd_inv < fd.shape[0])
    return tau_inv, fd_inv, pnts, th1, th2


def thth_map(CS, tau, fd, eta, edges, hermetian=True):
    """Nearest-bin CS -> theta-theta map (ththmod.py:56-116)."""
    tau_inv, fd_inv, pnts, th1, th2 = thth_index_maps(tau, fd, eta, edges)
    thth = np.zeros(tau_inv.shape, dtype=complex)
    thth[pnts] = CS[tau_inv[pnts], fd_inv[pnts]]
    thth *= np.sqrt(np.abs(2 * eta * (th2 - th1)))
    if hermetian:
        thth -= np.tril(thth)
        thth += np.conjugate(np.triu(thth).T)
        thth -= np.diag(np.diag(thth))
        thth -= np.diag(np.diag(thth[::-1, :]))[::-1, :]
        thth = np.nan_to_num(thth)
    return thth


def reduced_keep(tau, fd, eta, edges):
    """Boolean mask of the theta centres that survive thth_redmap's crop
    (ththmod.py:151-155), plus the centres themselves."""
    th_cents = theta_centres(edges)
    keep = ((th_cents**2) * eta < np.abs(tau.max())) * (
        np.abs(th_cents) < np.abs(fd.max()) / 2)
    return keep, th_cents


def reduced_edges(th_kept):
    """edges_red from the kept centres (ththmod.py:157-172)."""
    mid = (th_kept[:-1] + th_kept[1:]) / 2
    step = np.diff(mid).mean()
    return np.concatenate((np.array([mid[0] - step]), mid, np.array([mid[-1] + step])))


def thth_redmap(CS, tau, fd, eta, edges, hermetian=True):
    """theta-theta cropped to the square fully inside the CS (ththmod.py:119-173)."""
    thth = thth_map(CS, tau, fd, eta, edges, hermetian)
    keep, th_cents = reduced_keep(tau, fd, eta, edges)
    thth_red = thth[keep, :][:, keep]
    return thth_red, reduced_edges(th_cents[keep])


# --------------------------------------------------------------------------
# dominant eigenpair
# --------------------------------------------------------------------------
def Eval_calc(CS, tau, fd, eta, edges):
    """|largest-algebraic eigenvalue| of the reduced theta-theta (ththmod.py:371-401)."""
    thth_red, _ = thth_redmap(CS, tau, fd, eta, edges)
    v0 = np.copy(thth_red[thth_red.shape[0] // 2, :])
    v0 /= np.sqrt((np.abs(v0) ** 2).sum())
    w, V = eigsh(thth_red, 1, v0=v0, which="LA")
    return np.abs(w[0])


# --------------------------------------------------------------------------
# theta-theta -> CS (scatter) and the rank-1 model
# --------------------------------------------------------------------------
def rev_map(thth, tau, fd, eta, edges, hermetian=True):
    """Weighted-histogram inverse map theta-theta -> CS (ththmod.py:176-271)."""
    th_cents = theta_centres(edges)
    fd_map = th_cents[np.newaxis, :] - th_cents[:, np.newaxis]
    tau_map = eta * (th_cents[np.newaxis, :] ** 2 - th_cents[:, np.newaxis] ** 2)
    fd_edges = (np.linspace(0, fd.shape[0], fd.shape[0] + 1) - 0.5) * (fd[1] - fd[0]) + fd[0]
    tau_edges = (np.linspace(0, tau.shape[0], tau.shape[0] + 1) - 0.5) * (tau[1] - tau[0]) + tau[0]
    bins = (fd_edges, tau_edges)
    with np.errstate(divide="ignore", invalid="ignore"):
        wts = np.ravel(thth / np.sqrt(np.abs(2 * eta * fd_map.T)))
        x, y = np.ravel(fd_map), np.ravel(tau_map)
        recov = (np.histogram2d(x, y, bins=bins, weights=wts.real)[0]
                 + np.histogram2d(x, y, bins=bins, weights=wts.imag)[0] * 1j)
        norm = np.histogram2d(x, y, bins=bins)[0]
        if hermetian:
            recov += (np.histogram2d(-x, -y, bins=bins, weights=wts.real)[0]
                      - np.histogram2d(-x, -y, bins=bins, weights=wts.imag)[0] * 1j)
            norm += np.histogram2d(-x, -y, bins=bins)[0]
        recov /= norm
        recov = np.nan_to_num(recov)
    return recov.T


def modeler(CS, tau, fd, eta, edges):
    """Rank-1 theta-theta model mapped back to CS and to a model dynamic
    spectrum (ththmod.py:274-327, hermetian branch only -- the non-hermetian
    branch of the reference raises, ththmod.py:317-320)."""
    thth_red, edges_red = thth_redmap(CS, tau, fd, eta, edges)
    w, V = eigsh(thth_red, 1, which="LA")
    w = w[0]
    V = V[:, 0]
    thth2_red = np.outer(V, np.conjugate(V))
    thth2_red *= np.abs(w)
    recov = rev_map(thth2_red, tau, fd, eta, edges_red, hermetian=True)
    model = np.fft.ifft2(np.fft.ifftshift(recov)).real
    return thth_red, thth2_red, recov, model, edges_red, w, V


def chisq_calc(dspec, CS, tau, fd, eta, edges, N, mask=None):
    """chi**2 of the theta-theta model against the dynamic spectrum (ththmod.py:330-368)."""
    if mask is None:
        mask = np.isfinite(dspec)
    model = modeler(CS, tau, fd, eta, edges)[3][: dspec.shape[0], : dspec.shape[1]]
    return np.sum((model - dspec)[mask] ** 2) / N


# --------------------------------------------------------------------------
# eta sweep for one chunk
# --------------------------------------------------------------------------
def conjugate_spectrum(dspec, npad, tau=None, tau_mask=0.0):
    """Padded, shifted 2-D FFT of a chunk (ththmod.py:777-787)."""
    pad = np.pad(dspec, ((0, npad * dspec.shape[0]), (0, npad * dspec.shape[1])),
                 mode="constant", constant_values=dspec.mean())
    CS = np.fft.fftshift(np.fft.fft2(pad))
    if tau is not None:
        CS[np.abs(tau) < tau_mask] = 0
    return CS


def fit_eig_peak(etas, eigs, fw):
    """Parabola fit around the eigenvalue peak (ththmod.py:814-859).
    Returns (eta_fit, eta_sig, popt); NaNs when the fit fails."""
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


def single_search(dspec, freq, time, etas, edges, fw=0.1, npad=3, coher=True, tau_mask=0.0):
    """Curvature search of one chunk (ththmod.py:715-895) without plotting.
    Returns (eta_fit, eta_sig, mean freq, mean time, eigs)."""
    fd = fft_axis(time, 1000.0, npad)
    tau = fft_axis(freq, 1.0, npad)
    CS = conjugate_spectrum(dspec, npad, tau, tau_mask)
    src = CS if coher else np.abs(CS)
    eigs = np.zeros(etas.shape)
    for i in range(eigs.shape[0]):
        try:
            eigs[i] = Eval_calc(src, tau, fd, etas[i], edges)
        except Exception:
            eigs[i] = np.nan
    eta_fit, eta_sig, _ = fit_eig_peak(etas, eigs, fw)
    return eta_fit, eta_sig, freq.mean(), time.mean(), eigs


# --------------------------------------------------------------------------
# phase retrieval
# --------------------------------------------------------------------------
def single_chunk_retrieval(dspec2, edges, time, freq, eta, npad, tau_mask=0.0):
    """Wavefield of one chunk (ththmod.py:1390-1476), without the try/except."""
    fd = fft_axis(time, 1000.0, npad)
    tau = fft_axis(freq, 1.0, npad)
    CS = conjugate_spectrum(dspec2, npad, tau, tau_mask)
    thth_red, thth2_red, recov, model, edges_red, w, V = modeler(CS, tau, fd, eta, edges)
    ththE_red = thth_red * 0
    ththE_red[ththE_red.shape[0] // 2, :] = np.conjugate(V) * np.sqrt(w)
    recov_E = rev_map(ththE_red, tau, fd, eta, edges_red, hermetian=False)
    model_E = np.fft.ifft2(np.fft.ifftshift(recov_E))[: dspec2.shape[0], : dspec2.shape[1]]
    model_E *= dspec2.shape[0] * dspec2.shape[1] / 4
    return model_E


def mask_func(w):
    """ththmod.py:1479-1489."""
    x = np.linspace(0, w - 1, w)
    return np.sin((np.pi / 2) * x / w) ** 2


def mosaic(chunks):
    """ththmod.py:1492-1554."""
    ncf, nct, cwf, cwt = chunks.shape
    E_recov = np.zeros(((ncf - 1) * (cwf // 2) + cwf, (nct - 1) * (cwt // 2) + cwt), dtype=complex)
    for cf in range(ncf):
        for ct in range(nct):
            chunk_new = chunks[cf, ct, :, :]
            chunk_old = E_recov[cf * cwf // 2: cf * cwf // 2 + cwf, ct * cwt // 2: ct * cwt // 2 + cwt]
            mask = np.ones(chunk_new.shape)
            if cf > 0:
                mask[: cwf // 2, :] *= mask_func(cwf // 2)[:, np.newaxis]
            if cf < ncf - 1:
                mask[cwf // 2:, :] *= 1 - mask_func(cwf // 2)[:, np.newaxis]
            if ct > 0:
                mask[:, : cwt // 2] *= mask_func(cwt // 2)
            if ct < nct - 1:
                mask[:, cwt // 2:] *= 1 - mask_func(cwt // 2)
            rot = np.angle((chunk_old * np.conjugate(chunk_new) * mask).mean())
            E_recov[cf * cwf // 2: cf * cwf // 2 + cwf, ct * cwt // 2: ct * cwt // 2 + cwt] += (
                chunk_new * mask * np.exp(1j * rot))
    return E_recov


def gerchberg_saxton(wavefield, dyn, tau, niter=1):
    """Dynspec.gerchberg_saxton after calc_wavefield (dynspec.py:1858-1875)."""
    wavefield = np.array(wavefield, dtype=complex)
    d = dyn[: wavefield.shape[0], : wavefield.shape[1]]
    posdspec = np.isfinite(d) * (d > 0)
    wavefield *= np.sqrt(d[posdspec].mean() / np.abs(wavefield[posdspec] ** 2).mean())
    wavefield[posdspec] = np.sqrt(d[posdspec]) * np.exp(1j * np.angle(wavefield[posdspec]))
    for _ in range(niter):
        CWF = np.fft.fftshift(np.fft.fft2(wavefield))
        CWF[tau < 0] = 0
        wavefield = np.fft.ifft2(np.fft.ifftshift(CWF))
        wavefield[posdspec] = np.sqrt(d[posdspec]) * np.exp(1j * np.angle(wavefield[posdspec]))
    return wavefield


def calc_asymmetry(dspec2, edges, time, freq, eta, npad):
    """Arc asymmetry from the theta-theta eigenvector (ththmod.py:2385-2463), no try/except."""
    fd = fft_axis(time, 1000.0, npad)
    tau = fft_axis(freq, 1.0, npad)
    CS = conjugate_spectrum(dspec2, npad)
    thth_red, thth2_red, recov, model, edges_red, w, V = modeler(CS, tau, fd, eta, edges)
    cents = (edges_red[1:] + edges_red[:-1]) / 2
    leftV = V[: (cents.shape[0] - 1) // 2]
    rightV = V[1 + (cents.shape[0] - 1) // 2:]
    return (np.sum(np.abs(leftV) ** 2) - np.sum(np.abs(rightV) ** 2)) / (
        np.sum(np.abs(leftV) ** 2) + np.sum(np.abs(rightV) ** 2))
