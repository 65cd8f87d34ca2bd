"""Drop-in for the hot-path entry points of ``scintools.dynspec.Dynspec``.

Only what the hot path needs is here: a ``Dynspec`` that is built from an object
carrying the reference's attributes (a ``scint_sim.Simulation``, a ``BasicDyn``, a
reference ``Dynspec``, or plain arrays) with

* ``calc_sspec`` (dynspec.py:3584-3748) -> HIP kernels (``scint_sspec``);
* ``prep_thetatheta`` / ``thetatheta_single`` / ``fit_thetatheta``
  (dynspec.py:1348-1763): the chunked curvature search -- chunking, eta grids, edges and
  the global eta ~ nu**-2 fit stay on the host exactly as in the reference, every
  chunk's conjugate spectrum + eta sweep runs on the GPU (``ththmod.single_search``).

* ``scale_dyn(scale='lambda')``, ``norm_sspec``, ``fit_arc`` (dynspec.py:3928-3959,
  1920-2183, 970-1313): the arc-curvature search on the secondary spectrum that supplies
  ``prep_thetatheta`` with its default curvature bounds (``scintools_amd/arcfit.py``).

Cleaning, velocity / trapezoid rescaling and plotting are out of scope.
"""
import ctypes
import functools
import os
import warnings

import numpy as np
import scipy.constants as sc
import torch

from . import _lib, arcfit, psrflux, units
from . import ththmod as thth
from .device import DeviceBacked, empty, ptr, require_gpu, stream_ptr, to_device, workspace

_WINDOWS = {"hanning": np.hanning, "hamming": np.hamming,
            "blackman": np.blackman, "bartlett": np.bartlett}


def get_window(nt, nf, window="hanning", frac=0.1):
    """Edge tapers with a flat middle (scint_utils.py:810-832); host NumPy,
    the multiply is fused into the first FFT pass on the device."""
    try:
        fn = _WINDOWS[window.lower()]
    except KeyError:
        raise ValueError(f"Window unknown: {window!r}")
    cw = fn(np.floor(frac * nt))
    sw = fn(np.floor(frac * nf))
    chan_window = np.insert(cw, int(np.ceil(len(cw) / 2)), np.ones([nt - len(cw)]))
    subint_window = np.insert(sw, int(np.ceil(len(sw) / 2)), np.ones([nf - len(sw)]))
    return chan_window, subint_window


@functools.lru_cache(maxsize=32)
def _window_tables(nt, nf, window, frac, device):
    """The two taper vectors in HBM, kept per (shape, window, device): a repeated calc_sspec (the
    per-chunk loops of fit_thetatheta / the arc fits) re-uses them instead of rebuilding and
    uploading them on every call."""
    if window is None:
        return None, None
    cw, sw = get_window(nt, nf, window=window, frac=frac)
    return to_device(cw, torch.float64), to_device(sw, torch.float64)


@functools.lru_cache(maxsize=32)
def _postdark_tables(nrfft, ncfft, device):
    """Post-darkening factors of the prewhitened spectrum (dynspec.py:3700-3714), per device."""
    fd = np.array(list(range(int(-ncfft / 2), int(ncfft / 2))))
    td = np.array(list(range(0, int(nrfft / 2))))
    return (to_device(np.power(np.sin(np.multiply(sc.pi / ncfft, fd)), 2), torch.float64),
            to_device(np.power(np.sin(np.multiply(sc.pi / nrfft, td)), 2), torch.float64))


def sspec_device(dyn_t, prewhite=False, halve=True, window="hanning", window_frac=0.1):
    """Secondary spectrum of a device dynamic spectrum [nf, nt] float64 -> device
    tensor in dB, shape [(nrfft/2 if halve else nrfft), ncfft] (dynspec.py:3665-3721)."""
    lib = _lib.load()
    require_gpu()
    nf, nt = (int(v) for v in dyn_t.shape)
    nrfft = int(2 ** (np.ceil(np.log2(nf)) + 1))      # dynspec.py:3677
    ncfft = int(2 ** (np.ceil(np.log2(nt)) + 1))      # dynspec.py:3678
    if prewhite and not halve:
        raise RuntimeError("Cannot apply prewhite to full frame")   # dynspec.py:3717
    wt, wf = _window_tables(nt, nf, window, window_frac, torch.cuda.current_device())
    pd_fd, pd_td = _postdark_tables(nrfft, ncfft, torch.cuda.current_device()) if prewhite else (None, None)
    need = ctypes.c_size_t()
    _lib.check(lib.scint_sspec_workspace_bytes(nf, nt, ctypes.byref(need)), "sspec_workspace_bytes")
    ws = workspace.get(need.value)
    out = empty((nrfft // 2 if halve else nrfft, ncfft), torch.float64)
    rc = lib.scint_sspec(ptr(dyn_t), nf, nt, ptr(wt), ptr(wf), 1 if prewhite else 0,
                         1 if halve else 0, ptr(pd_fd), ptr(pd_td), ptr(out), ptr(ws), ws.numel(),
                         stream_ptr())
    _lib.check(rc, "scint_sspec")
    return out


class Dynspec:
    """Dynamic-spectrum holder with the reference's attribute names
    (dynspec.py:400-413) and a GPU ``calc_sspec``.

    ``sspec``, ``lamsspec`` and ``lamdyn`` are the reference's NumPy attributes, but a freshly
    computed one stays in HBM until it is first read (``device.DeviceBacked``): the chain
    ``scale_dyn -> calc_sspec(lamsteps=True) -> fit_arc`` then never crosses PCIe."""
    sspec = DeviceBacked("sspec")
    lamsspec = DeviceBacked("lamsspec")
    chunks = DeviceBacked("chunks")          # the retrieved wavefield chunks (1 GB for a 4096^2 observation): in HBM until read
    wavefield = DeviceBacked("wavefield")
    lamdyn = DeviceBacked("lamdyn")

    def __init__(self, filename=None, dyn=None, verbose=True, process=False, lamsteps=False,
                 remove_short_subs=True, subint_thresh=2.33, mjd=None):
        if process:
            raise NotImplementedError("process=True: cleaning is outside the accelerated hot path "
                                      "(and calls a missing method in the reference, dynspec.py:416)")
        if filename:
            self.load_file(filename, verbose=verbose, lamsteps=lamsteps, subint_thresh=subint_thresh,
                           remove_short_subs=remove_short_subs, mjd=mjd)
        elif dyn is not None:
            self.load_dyn_obj(dyn, verbose=verbose, lamsteps=lamsteps)
        else:
            raise ValueError("Error: No dynamic spectrum file or object")

    # ------------------------------------------------------------------ psrflux text I/O (host)
    def load_file(self, filename, verbose=True, process=False, lamsteps=False, remove_short_subs=True,
                  subint_thresh=2.33, mjd=None):
        """Load a psrflux-format dynamic spectrum (reference: dynspec.py:144-230).  The parsing
        lives in :mod:`scintools_amd.psrflux` (single-pass tokeniser, optional binary side-car);
        the attributes set here are the reference's."""
        if process:
            raise NotImplementedError("process=True: cleaning is outside the accelerated hot path")
        if verbose:
            print("LOADING {0}...".format(filename))
        parsed = psrflux.load_sidecar(filename) or psrflux.read_table(filename)
        obs = psrflux.observation(*parsed, mjd=mjd, mjd_known=getattr(self, "mjd", None))
        self.name = os.path.basename(filename)
        self.filename = filename
        self.__dict__.update(obs)
        if remove_short_subs and np.std(np.diff(self.times)) != 0:
            self.remove_short_subs(threshold=subint_thresh)
        self.lamsteps = lamsteps

    def remove_short_subs(self, threshold=2.33):
        """Drop short sub-integrations at the start of the observation (reference:
        dynspec.py:232-258) and re-derive the time attributes."""
        k = psrflux.leading_short_subs(self.times, threshold)
        self.dyn = self.dyn[:, k:]
        self.times = self.times[k:]
        self.mjd += self.times[0] / 86400
        self.times = self.times - self.times[0]
        self.nsub = len(self.times)
        self.dt = round(np.mean(np.diff(self.times)), 3)
        self.tobs = round(max(self.times) + self.dt, 3)

    def write_file(self, filename=None, verbose=True, note=None, sidecar=False):
        """Write the dynamic spectrum in psrflux format (reference: dynspec.py:330-376); with
        ``sidecar=True`` also the binary image that ``load_file`` prefers while it is current."""
        if filename is None:
            stem, ext = self.filename.rsplit('.', 1)
            filename = f"{stem}.processed.{ext}"
        psrflux.write(filename, self.header, self.mjd, self.times, self.freqs, self.dyn, note=note)
        if sidecar:
            psrflux.save_sidecar(filename, *psrflux.read_table(filename))
        if verbose:
            print("Wrote dynamic spectrum file as {}".format(filename))

    def load_dyn_obj(self, dyn, verbose=True, process=False, lamsteps=False):
        """Copy the reference's attribute set from any object that has it (dynspec.py:378-419)."""
        self.name = getattr(dyn, "name", "dynspec")
        self.header = getattr(dyn, "header", [self.name])
        self.times = np.asarray(dyn.times, dtype=float)
        self.freqs = np.asarray(dyn.freqs, dtype=float)
        self.dyn = np.asarray(dyn.dyn, dtype=float)
        self.nchan = int(getattr(dyn, "nchan", self.dyn.shape[0]))
        self.nsub = int(getattr(dyn, "nsub", self.dyn.shape[1]))
        self.df = float(getattr(dyn, "df", np.abs(self.freqs[1] - self.freqs[0])))
        self.dt = float(getattr(dyn, "dt", self.times[1] - self.times[0]))
        self.bw = float(getattr(dyn, "bw", np.ptp(self.freqs) + self.df))
        self.freq = float(getattr(dyn, "freq", np.mean(self.freqs)))
        tobs = getattr(dyn, "tobs", None)
        self.tobs = float(tobs) if tobs is not None else float(np.ptp(self.times) + self.dt)
        m = getattr(dyn, "mjd", None)
        self.mjd = m if m is not None else 60000.0
        self.lamsteps = lamsteps
        if verbose:
            print(f"LOADING DYNSPEC OBJECT {self.name}...")

    def calc_sspec(self, prewhite=False, halve=True, plot=False, lamsteps=False, input_dyn=None,
                   input_x=None, input_y=None, trap=False, window="hanning", window_frac=0.1,
                   return_sspec=False, velocity=False):
        """Secondary spectrum (dynspec.py:3584-3748) on the GPU.

        Sets ``self.sspec / self.fdop / self.tdel`` (``self.lamsspec`` and ``self.beta`` with
        ``lamsteps``, on the wavelength-scaled spectrum of ``scale_dyn``) or, with ``input_dyn``
        or ``return_sspec``, returns ``(fdop, yaxis, sec)``.  The velocity- and
        trapezoid-rescaled variants are outside the hot path and raise ``NotImplementedError``.
        """
        if velocity or trap:
            raise NotImplementedError("velocity / trap need Dynspec.scale_dyn('velocity'/'trapezoid'), "
                                      "which is outside the accelerated hot path")
        if plot:
            raise NotImplementedError("plotting is outside the accelerated hot path")
        cls = type(self)
        if input_dyn is not None:
            dyn_t = to_device(input_dyn, torch.float64)
        elif lamsteps:
            if not cls.lamdyn.present(self):
                self.scale_dyn()
            dyn_t = cls.lamdyn.tensor(self)
        else:
            dyn_t = to_device(self.dyn, torch.float64)
        sec_t = sspec_device(dyn_t, prewhite=prewhite, halve=halve, window=window, window_frac=window_frac)
        nf, nt = dyn_t.shape
        nrfft = int(2 ** (np.ceil(np.log2(nf)) + 1))
        ncfft = int(2 ** (np.ceil(np.log2(nt)) + 1))
        td = np.array(list(range(0, int(nrfft / 2) if halve else int(nrfft))))
        fd = np.array(list(range(int(-ncfft / 2), int(ncfft / 2))))
        fdop = np.reshape(np.multiply(fd, 1e3 / (ncfft * self.dt)), [len(fd)])   # mHz
        tdel = np.reshape(np.divide(td, (nrfft * self.df)), [len(td)])           # us
        if lamsteps:
            beta = np.divide(td, (nrfft * self.dlam))                            # m^-1 (dynspec.py:3703-3704)
        if input_dyn is None and not return_sspec:
            if lamsteps:
                cls.lamsspec.park(self, sec_t)      # copied to the host when first read
                self.beta = beta
            else:
                cls.sspec.park(self, sec_t)
            self.fdop = fdop
            self.tdel = tdel
            return None
        return fdop, (beta if lamsteps else tdel), sec_t.cpu().numpy()

    # ------------------------------------------------------------------ arc normalisation
    def scale_dyn(self, scale='lambda', window_frac=0.1, pars=None, parfile=None, window='hanning',
                  spacing='auto', s=None, d=None, vism_ra=None, vism_dec=None, Omega=None, inc=None,
                  vism_zeta=None, zeta=None, lamsteps=False, velocity=False, trap=False):
        """Rescale the dynamic spectrum (dynspec.py:3872-4080): the equal-wavelength resample
        (``'lambda'`` / ``'wavelength'`` / ``lamsteps``) runs on the GPU -- one not-a-knot cubic
        spline per time column -- and sets ``lamdyn / lam / nlam / dlam``.  The velocity, orbit and
        trapezoid scalings need ephemerides and are outside the accelerated hot path."""
        if ('velocity' in scale) or ('orbit' in scale) or velocity or ('trapezoid' in scale) or trap:
            raise NotImplementedError("velocity / orbit / trapezoid scaling is outside the accelerated hot path")
        if ('lambda' in scale) or ('wavelength' in scale) or lamsteps:
            arcfit.scale_dyn_lambda(self, spacing=spacing)

    norm_sspec = arcfit.norm_sspec
    fit_arc = arcfit.fit_arc

    @property
    def normsspec(self):
        """2-D normalised secondary spectrum of the last ``norm_sspec`` (masked array); copied
        from the device on first access."""
        return arcfit.normsspec_host(self)

    @normsspec.setter
    def normsspec(self, value):
        self._normsspec_host = value

    @property
    def mask(self):
        """Mask of ``normsspec`` (dynspec.py:2118-2126)."""
        override = getattr(self, "_mask_host", None)
        return override if override is not None else np.ma.getmaskarray(arcfit.normsspec_host(self))

    @mask.setter
    def mask(self, value):
        self._mask_host = value

    def calc_acf(self, method='direct', input_dyn=None, normalise=True, window_frac=0.1):
        """Autocovariance function (dynspec.py:3750-3814), 'direct' method on the GPU:
        zero-padded fft2 -> |.|^2 -> ifft2 -> fftshift -> real (-> / max).  Sets ``self.acf`` or,
        with ``input_dyn``, returns it.  Non-finite pixels are not supported here (the reference
        only excludes them from the mean); method='sspec' is outside the hot path."""
        if method != 'direct':
            raise NotImplementedError("calc_acf(method='sspec') is outside the accelerated hot path")
        lib = _lib.load()
        require_gpu()
        dyn = self.dyn if input_dyn is None else np.asarray(input_dyn)
        if not np.all(np.isfinite(dyn)):
            raise ValueError("calc_acf on the GPU needs a finite dynamic spectrum (run refill first)")
        dyn_t = to_device(dyn, torch.float64)
        nf, nt = (int(v) for v in dyn_t.shape)
        need = ctypes.c_size_t()
        _lib.check(lib.scint_acf_workspace_bytes(nf, nt, ctypes.byref(need)), "acf_workspace_bytes")
        ws = workspace.get(need.value)
        out = empty((2 * nf, 2 * nt), torch.float64)
        # with input_dyn the reference does NOT subtract the mean (dynspec.py:3786-3789)
        rc = lib.scint_acf(ptr(dyn_t), nf, nt, 1 if input_dyn is None else 0, 1 if normalise else 0, ptr(out),
                           ptr(ws), ws.numel(), stream_ptr())
        _lib.check(rc, "scint_acf")
        arr = out.cpu().numpy()
        if input_dyn is None:
            self.acf = arr
            return None
        return arr

    # ------------------------------------------------------------------ theta-theta
    def prep_thetatheta(self, fw=.1, npad=3, verbose=False, fitting_proc='standard', **kwargs):
        """Set the theta-theta search parameters (dynspec.py:1348-1537).

        Same keywords as the reference: cwf, cwt, fref, eta_min, eta_max, nedge, edges_lim,
        tau_lim, tau_mask (bare numbers in s**3 / mHz / us / MHz, or astropy Quantities).
        Without ``eta_min`` / ``eta_max`` the bounds come from ``fit_arc(lamsteps=True)`` on
        the secondary spectrum, as in the reference (dynspec.py:1458-1473).  The 'thin'
        procedure (rectangular theta-theta + SVD) is out of scope.
        """
        fitting_procs = ['standard', 'thin', 'incoherent']
        assert fitting_proc in fitting_procs, f'fitting_proc must be one of {fitting_procs}'
        if fitting_proc == 'thin':
            raise NotImplementedError("fitting_proc='thin' (two-curvature SVD) is outside the hot path")
        val = lambda key, unit, name: float(units.strip(kwargs[key], name, unit, warn=False))
        self.thetatheta_proc = fitting_proc
        self.npad = npad
        self.fw = fw
        if 'cwf' in kwargs:
            self.cwf = 2 * (kwargs['cwf'] // 2)
            self.ncf_fit = self.dyn.shape[0] // self.cwf
            self.ncf_ret = (self.dyn.shape[0] // (self.cwf // 2)) - 1
        else:
            self.cwf, self.ncf_fit, self.ncf_ret = self.dyn.shape[0], 1, 1
        if 'cwt' in kwargs:
            self.cwt = 2 * (kwargs['cwt'] // 2)
            self.nct_fit = self.dyn.shape[1] // self.cwt
            self.nct_ret = (self.dyn.shape[1] // (self.cwt // 2)) - 1
        else:
            self.cwt, self.nct_fit, self.nct_ret = self.dyn.shape[1], 1, 1
        tau_lim = val('tau_lim', 'us', 'Tau Limit') if 'tau_lim' in kwargs else None
        self.fref = val('fref', 'MHz', 'reference frequency') if 'fref' in kwargs else self.freqs.mean()

        fd = thth.fft_axis(self.times[:self.cwt], 1000.0)       # mHz   (dynspec.py:1445)
        tau = thth.fft_axis(self.freqs[:self.cwf], 1.0)         # us    (dynspec.py:1446)
        eta_min = 4 * (tau[1] - tau[0]) / fd.max()**2            # dynspec.py:1448-1451
        eta_max = tau.max() / (fd[1] - fd[0])**2
        eta_min *= (self.freqs.max() / self.fref)**2
        eta_max *= (self.freqs.min() / self.fref)**2
        self.eta_min = max((val('eta_min', 's3', 'eta_min'), eta_min)) if 'eta_min' in kwargs else eta_min
        self.eta_max = min((val('eta_max', 's3', 'eta_max'), eta_max)) if 'eta_max' in kwargs else eta_max
        if not ('eta_min' in kwargs and 'eta_max' in kwargs):         # dynspec.py:1458-1473
            # curvature from the secondary spectrum; s**3 <-> 1/(m mHz**2) through c / fref**2
            if not hasattr(self, "betaeta"):
                self.fit_arc(lamsteps=True, numsteps=1e4, etamin=self.eta_min * self.fref**2 * 1e6 / sc.c,
                             etamax=self.eta_max * self.fref**2 * 1e6 / sc.c, delmax=tau_lim)
            eta_hough = sc.c * self.betaeta / (self.fref**2 * 1e6)
            err_hough = sc.c * 2 * max((self.betaetaerr, self.betaetaerr2)) / (self.fref**2 * 1e6)
            if 'eta_min' not in kwargs:
                self.eta_min = max((self.eta_min, eta_hough - err_hough))
            if 'eta_max' not in kwargs:
                self.eta_max = min((self.eta_max, eta_hough + err_hough))
        l0, l1 = np.log10(self.eta_min), np.log10(self.eta_max)
        self.neta = int(1 + (l1 - l0) / np.log10(1 + self.fw / 10))     # dynspec.py:1478
        fd_cut = (fd.max() / 2) * (self.fref / self.freqs.max())
        edges_lim = min((val('edges_lim', 'mHz', 'edges limit'), fd_cut)) if 'edges_lim' in kwargs else fd_cut
        if tau_lim is not None:
            edges_lim = min((edges_lim, np.sqrt(tau_lim / self.eta_max)))
        if 'nedge' in kwargs:
            assert np.mod(kwargs['nedge'], 2) == 0, 'nedge must be even!'
            self.edges = np.linspace(-edges_lim, edges_lim, kwargs['nedge'])
        else:                                                     # dynspec.py:1500-1503
            self.edges = np.asarray(thth.min_edges(edges_lim, fd, tau,
                                                   self.eta_max * (self.fref / self.freqs.min()), 2)) \
                * (self.freqs.min() / self.fref)
            if units.HAVE_ASTROPY:
                self.edges = np.asarray(getattr(self.edges, "value", self.edges))
        self.thth_tau_mask = val('tau_mask', 'us', 'tau_mask') if 'tau_mask' in kwargs else 0.0
        if verbose:
            print("\n\t THETA-THETA PROPERTIES\n")
            print(f'Channels per chunk: {self.cwf}')
            print(f'Time bins per chunk: {self.cwt}')
            print(f'Number of fitting chunks: {self.ncf_fit}x{self.nct_fit}')
            print(f'Reference Frequency: {self.fref} MHz')
            print(f'Eta range: {self.eta_min} to {self.eta_max} s3 with {self.neta} points')
            print(f'Edges has {self.edges.shape[0]} point out to {self.edges[-1]} mHz')
            print(f'Fractional fitting width: {self.fw}')
            print(f'Zero paddings: {self.npad}')
            print(f'Fitting Procedure: {self.thetatheta_proc}')
            print(f'Masking |tau| < {self.thth_tau_mask} us')

    def _chunk(self, cf, ct):
        fs = slice(cf * self.cwf, (cf + 1) * self.cwf)
        ts = slice(ct * self.cwt, (ct + 1) * self.cwt)
        return fs, ts

    def _chunk_etas(self, freq2):
        return np.logspace(np.log10(self.eta_min), np.log10(self.eta_max), self.neta) \
            * (self.fref / freq2.mean())**2

    def thetatheta_single(self, cf=0, ct=0, fname=None, verbose=False, plot=True, arrays=False):
        """theta-theta curvature search on one chunk (dynspec.py:1539-1655).  Same defaults as
        the reference: with ``arrays=True`` returns (etas, eigs, popt); the diagnostic plot is
        outside the accelerated path, so ``plot=True`` only warns."""
        if not hasattr(self, 'cwf'):
            self.prep_thetatheta(verbose=verbose)            # dynspec.py:1555-1556
        if plot:
            warnings.warn("scintools_amd: thetatheta_single does not draw the diagnostic plot; "
                          "pass arrays=True for the curve and the fit")
        cf, ct = min(cf, self.ncf_fit - 1), min(ct, self.nct_fit - 1)
        fs, ts = self._chunk(cf, ct)
        time2, freq2 = self.times[ts], self.freqs[fs]
        tau = thth.fft_axis(freq2, 1.0, self.npad)
        fd = thth.fft_axis(time2, 1000.0, self.npad)
        dspec2 = np.copy(self.dyn[fs, ts])
        dspec2 -= np.nanmean(dspec2)
        cs = thth.conjugate_spectrum(np.nan_to_num(dspec2), self.npad, tau, self.thth_tau_mask,
                                     self.thetatheta_proc != 'incoherent', pad_value=0.0)
        etas = self._chunk_etas(freq2)
        edges = self.edges * (freq2.mean() / self.fref)
        eigs = thth.eval_sweep(cs, tau, fd, etas, edges)
        _, _, popt = thth.fit_eig_peak(etas, eigs, self.fw)
        if arrays:
            return etas, eigs, popt

    def _fit_chunks(self, group_all):
        """Curvature fits of the listed chunks on this GPU: rows [eta_fit, eta_sig, eigs...].
        Chunks are grouped so that a stack of conjugate spectra stays below ~8 GiB; each group is
        one ``eval_sweep_multi`` call (all (chunk, eta) pairs batched together)."""
        coher = (self.thetatheta_proc != 'incoherent')
        R, C = (self.npad + 1) * self.cwf, (self.npad + 1) * self.cwt
        per_group = max(1, int((8 << 30) // (16 * R * C)))
        rows = np.full((len(group_all), 2 + self.neta), np.nan)
        for g0 in range(0, len(group_all), per_group):
            group = group_all[g0:g0 + per_group]
            stack = empty((len(group), R, C), torch.complex128)
            grids, etas_list, pads = [], [], []
            # the group's chunks travel to the device in ONE array (an upload from pageable memory per chunk, and a device mean
            # read back per chunk for the padding value, block the host once each: ththmod.chunk_retrieval_batch)
            d_all = np.empty((len(group), self.cwf, self.cwt))
            for k, (cf, ct) in enumerate(group):
                dspec2, freq2, time2, etas, edges = self._search_params(cf, ct)[:5]
                fd = thth.fft_axis(time2, 1000.0, self.npad)          # ththmod.py:773
                tau = thth.fft_axis(freq2, 1.0, self.npad)            # ththmod.py:774
                d_all[k] = dspec2
                pads.append(float(d_all[k].mean()))                   # the padding value (ththmod.py:783)
                grids.append((tau, fd, edges))
                etas_list.append(etas)
            d_t = thth.to_device(d_all, torch.float64)
            for k in range(len(group)):
                thth.conjugate_spectrum(d_t[k], self.npad, grids[k][0], self.thth_tau_mask, coher, pad_value=pads[k], out=stack[k])
            eig_list = thth.eval_sweep_multi(stack, grids, etas_list)
            for k, (etas, eigs) in enumerate(zip(etas_list, eig_list)):
                eta_fit, eta_sig, _ = thth.fit_eig_peak(etas, eigs, self.fw)   # ththmod.py:814-859
                rows[g0 + k, 0], rows[g0 + k, 1] = eta_fit, eta_sig
                rows[g0 + k, 2:] = eigs
        return rows

    def _search_params(self, cf, ct, verbose=False):
        """The 12-element parameter list the reference hands to ``single_search`` for fitting
        chunk (cf, ct) (dynspec.py:1689-1704), plain floats."""
        fs, ts = self._chunk(cf, ct)
        freq2, time2 = np.copy(self.freqs[fs]), np.copy(self.times[ts])
        dspec2 = np.copy(self.dyn[fs, ts])
        dspec2 -= np.nanmean(dspec2)
        dspec2 = np.nan_to_num(dspec2)
        coher = (self.thetatheta_proc != 'incoherent')
        return [dspec2, freq2, time2, self._chunk_etas(freq2), self.edges * (freq2.mean() / self.fref),
                None, False, self.fw, self.npad, coher, self.thth_tau_mask, verbose]

    def fit_thetatheta(self, verbose=False, plot=False, pool=None, time_avg=False):
        """Curvature search over all fitting chunks and the global eta ~ nu**-2 fit
        (dynspec.py:1657-1763).  Sets eta_evo, eta_evo_err, f0s, t0s, ththeta, ththetaerr
        (plain floats: s**3, MHz, s).

        Where the chunks run:
          * ``pool=None`` (default): every chunk's conjugate spectrum goes into one device stack
            and all (chunk, eta) pairs run as ONE continuously batched sweep on this GPU; under an
            initialised ``torch.distributed`` group the chunks are dealt round-robin to the ranks
            (one GPU each) and the fitted curvatures are all-gathered (``sweep.sharded_chunks``);
          * ``pool`` given: exactly the reference's ``pool.map(thth.single_search, pars)``
            (dynspec.py:1715-1719) -- with ``sweep.gpu_pool(n)`` every worker process drives its
            own GPU.
        """
        if not hasattr(self, 'cwf'):
            self.prep_thetatheta(verbose=verbose)             # dynspec.py:1673-1674
        if plot:
            warnings.warn("scintools_amd: fit_thetatheta does not plot the curvature evolution")
        self.eta_evo = np.zeros((self.ncf_fit, self.nct_fit))
        self.eta_evo_err = np.zeros((self.ncf_fit, self.nct_fit))
        self.f0s = np.zeros(self.ncf_fit)
        self.t0s = np.zeros(self.nct_fit)
        self.thth_eigs = np.zeros((self.ncf_fit, self.nct_fit, self.neta))
        chunks = [(cf, ct) for cf in range(self.ncf_fit) for ct in range(self.nct_fit)]
        for cf, ct in chunks:
            fs, ts = self._chunk(cf, ct)
            self.f0s[cf] = self.freqs[fs].mean()
            self.t0s[ct] = self.times[ts].mean()
        if pool is not None:
            res = pool.map(thth.single_search, [self._search_params(cf, ct, verbose) for cf, ct in chunks])
            for (cf, ct), r in zip(chunks, res):
                # with astropy present single_search returns Quantities (s**3): strip before the
                # plain-float arrays take them (a NumPy scalar assignment of a Quantity raises)
                self.eta_evo[cf, ct] = float(units.strip(r[0], 'eta', 's3', warn=False))
                self.eta_evo_err[cf, ct] = float(units.strip(r[1], 'eta_sig', 's3', warn=False))
                # single_search drops failed curvatures from the curve it returns (ththmod.py:817):
                # only a complete curve can be put back on the eta grid
                curve = np.asarray(r[4], dtype=float)
                self.thth_eigs[cf, ct] = curve if curve.shape == (self.neta,) else np.nan
        else:
            from . import sweep
            fits = sweep.sharded_chunks(len(chunks), lambda idx: self._fit_chunks([chunks[i] for i in idx]),
                                        self.neta)
            for (cf, ct), row in zip(chunks, fits):
                self.eta_evo[cf, ct], self.eta_evo_err[cf, ct] = row[0], row[1]
                self.thth_eigs[cf, ct] = row[2:]
        f0 = self.f0s[:, np.newaxis]
        with np.errstate(divide='ignore', invalid='ignore'):
            if time_avg:                                                   # dynspec.py:1724-1732
                eta_avg = np.nanmean(self.eta_evo, 1)
                eta_count = np.nansum(self.eta_evo, 1) / eta_avg
                avg_err = np.nanstd(self.eta_evo, 1) / np.sqrt(eta_count - 1)
                tofit = np.isfinite(eta_avg) * np.isfinite(avg_err)
                A = (np.sum(eta_avg[tofit] / (self.f0s * avg_err)[tofit] ** 2) /
                     np.sum(1 / (self.f0s**2 * avg_err)[tofit] ** 2))
                A_err = np.sqrt(1 / np.sum(2 / ((self.f0s**2) * avg_err)[tofit] ** 2))
            else:                                                          # dynspec.py:1734-1742
                tofit = np.isfinite(self.eta_evo) * np.isfinite(self.eta_evo_err)
                A = (np.sum(self.eta_evo[tofit] / (f0 * self.eta_evo_err)[tofit] ** 2) /
                     np.sum(1 / ((f0**2) * self.eta_evo_err)[tofit] ** 2))
                A_err = np.sqrt(1 / np.sum(2 / ((f0**2) * self.eta_evo_err)[tofit] ** 2))
        self.ththeta = A / self.fref**2
        self.ththetaerr = A_err / self.fref**2

    # ------------------------------------------------------------------ phase retrieval
    def thetatheta_chunks(self, verbose=False, pool=None, memmap=False):
        """theta-theta phase retrieval on all half-overlapping chunks (dynspec.py:1765-1826):
        fills ``self.chunks[ncf_ret, nct_ret, cwf, cwt]``.

        ``pool=None`` (default): every chunk's conjugate spectrum goes into one device stack, the dominant
        eigenpairs of all chunks come from ONE batched sweep and the back-maps / inverse FFTs are queued
        without a host round trip in between (``ththmod.chunk_retrieval_batch``).  With a ``pool`` it is the
        reference's ``pool.map(thth.single_chunk_retrieval, pars)`` (dynspec.py:1806-1812); `memmap` is
        accepted for signature compatibility."""
        if not hasattr(self, "ththeta"):
            self.fit_thetatheta(verbose=verbose, pool=pool)
        if pool is not None:
            self.chunks = np.zeros((self.ncf_ret, self.nct_ret, self.cwf, self.cwt), dtype=complex)
            pars = []
            for cf in range(self.ncf_ret):
                fs = slice(cf * (self.cwf // 2), cf * (self.cwf // 2) + self.cwf)
                freq2 = np.copy(self.freqs[fs])
                freq = freq2.mean()
                eta = self.ththeta * (self.fref / freq)**2
                for ct in range(self.nct_ret):
                    ts = slice(ct * (self.cwt // 2), ct * (self.cwt // 2) + self.cwt)
                    time2 = np.copy(self.times[ts])
                    dspec2 = np.copy(self.dyn[fs, ts])
                    dspec2 -= np.nanmean(dspec2)
                    dspec2 = np.nan_to_num(dspec2)
                    pars.append((dspec2, self.edges * (freq / self.fref), time2, freq2, eta, ct, cf, self.npad,
                                 self.thth_tau_mask, verbose))
            for res in pool.map(thth.single_chunk_retrieval, pars):
                self.chunks[res[1], res[2], :, :] = res[0]
            return
        # Round 6: the chunks are cut out of the dynamic spectrum ON the device (one upload of `dyn`; window, nanmean, nan_to_num and
        # the padding value in NumPy's own summation order: ththmod.chunk_cut_device) and the retrieved chunks stay there for the
        # mosaic -- the host loop above copied 0.5 GB up and 1 GB down and spent 0.2 s in nanmean / nan_to_num of 961 windows.
        pars, origins = [], []
        for cf in range(self.ncf_ret):
            fs = slice(cf * (self.cwf // 2), cf * (self.cwf // 2) + self.cwf)
            freq2 = np.copy(self.freqs[fs])
            freq = freq2.mean()
            eta = self.ththeta * (self.fref / freq)**2
            for ct in range(self.nct_ret):
                ts = slice(ct * (self.cwt // 2), ct * (self.cwt // 2) + self.cwt)
                pars.append((None, self.edges * (freq / self.fref), np.copy(self.times[ts]), freq2, eta))
                origins.append((cf * (self.cwf // 2), ct * (self.cwt // 2)))
        dyn_h = np.asarray(self.dyn, dtype=float)
        dyn_t = thth.to_device(dyn_h, torch.float64)
        d_t, pad_t = thth.chunk_cut_device(dyn_t, origins, self.cwf, self.cwt,
                                           fortran_order=dyn_h.flags.f_contiguous and not dyn_h.flags.c_contiguous)
        out_t = thth.chunk_retrieval_batch(pars, self.npad, self.thth_tau_mask, verbose=verbose, dev_chunks=d_t,
                                           dev_pads=pad_t.cpu().numpy(), out_device=True)
        type(self).chunks.park(self, out_t.reshape(self.ncf_ret, self.nct_ret, self.cwf, self.cwt))

    def calc_wavefield(self, verbose=False, pool=None, gs=False, memmap=False, niter=1):
        """Mosaic the chunks into the final wavefield (dynspec.py:1828-1856)."""
        cls = type(self)
        if not cls.chunks.present(self):
            self.thetatheta_chunks(verbose=verbose, pool=pool, memmap=memmap)
        # the mosaic on the device (ththmod.mosaic_device: the reference's loop, NumPy's summation order, bit-identical to
        # thth.mosaic on the same chunks); the wavefield stays in HBM until it is read
        cls.wavefield.park(self, thth.mosaic_device(cls.chunks.tensor(self, torch.complex128)))
        if gs:
            self.gerchberg_saxton(verbose=verbose, pool=pool, niter=niter)

    def gerchberg_saxton(self, niter=1, verbose=False, pool=None):
        """Gerchberg-Saxton: enforce the measured amplitudes and causality (tau >= 0)
        (dynspec.py:1858-1875); the FFT <-> projection iterations run on the GPU."""
        self.calc_wavefield(verbose=verbose, pool=pool)
        F = self.wavefield.shape[0]
        tau = thth.fft_axis(self.freqs[:F], 1.0)
        self.wavefield = thth.gerchberg_saxton_device(self.wavefield, self.dyn, tau, niter=niter)

    def calc_asymmetry(self, verbose=False, pool=None):
        """Arc asymmetry of every fitting chunk (dynspec.py:1877-1906), including the
        reference's time slice ``ct*cwt//2 : (ct+1)*cwt``."""
        if not hasattr(self, "ththeta"):
            self.fit_thetatheta(verbose=verbose, pool=pool)
        self.asymmetry = np.zeros((self.ncf_fit, self.nct_fit), dtype=complex)
        for cf in range(self.ncf_fit):
            fs = slice(cf * self.cwf, (cf + 1) * self.cwf)
            freq2 = np.copy(self.freqs[fs])
            freq = freq2.mean()
            eta = self.ththeta * (self.fref / freq)**2
            for ct in range(self.nct_fit):
                ts = slice(ct * self.cwt // 2, (ct + 1) * self.cwt)
                time2 = np.copy(self.times[ts])
                dspec2 = np.copy(self.dyn[fs, ts])
                dspec2 -= np.nanmean(dspec2)
                dspec2 = np.nan_to_num(dspec2)
                params = (dspec2, self.edges * (freq / self.fref), time2, freq2, eta, ct, cf, self.npad, verbose)
                self.asymmetry[cf, ct] = thth.calc_asymmetry(params)[0]
