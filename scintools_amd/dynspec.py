"""Drop-in for the secondary-spectrum entry point of ``scintools.dynspec.Dynspec``.

Only what the hot path needs is here: a ``Dynspec`` that is built from an object
carrying the reference's attributes (a ``scint_sim.Simulation``, a reference
``Dynspec``, or plain arrays) and its ``calc_sspec`` with the reference's
signature (dynspec.py:3584-3748) running as HIP kernels (``scint_sspec``).  File
I/O, cleaning, fitting and plotting are out of scope.
"""
import ctypes

import numpy as np
import scipy.constants as sc
import torch

from . import _lib
from .device import empty, ptr, require_gpu, stream_ptr, to_device, workspace

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
    wt = wf = pd_fd = pd_td = None
    if window is not None:
        cw, sw = get_window(nt, nf, window=window, frac=window_frac)
        wt, wf = to_device(cw, torch.float64), to_device(sw, torch.float64)
    if prewhite:
        fd = np.array(list(range(int(-ncfft / 2), int(ncfft / 2))))
        td = np.array(list(range(0, int(nrfft / 2))))
        pd_fd = to_device(np.power(np.sin(np.multiply(sc.pi / ncfft, fd)), 2), torch.float64)
        pd_td = to_device(np.power(np.sin(np.multiply(sc.pi / nrfft, td)), 2), torch.float64)
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
    (dynspec.py:400-413) and a GPU ``calc_sspec``."""

    def __init__(self, filename=None, dyn=None, verbose=True, process=False, lamsteps=False,
                 remove_short_subs=True, subint_thresh=2.33, mjd=None):
        if filename:
            raise NotImplementedError("psrflux file I/O is outside the accelerated hot path")
        if dyn is None:
            raise ValueError("Error: No dynamic spectrum file or object")
        if process:
            raise NotImplementedError("process=True: cleaning is outside the accelerated hot path "
                                      "(and calls a missing method in the reference, dynspec.py:416)")
        self.load_dyn_obj(dyn, verbose=verbose, lamsteps=lamsteps)

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

        Sets ``self.sspec / self.fdop / self.tdel`` or, with ``input_dyn`` or
        ``return_sspec``, returns ``(fdop, tdel, sec)``.  The wavelength-,
        velocity- and trapezoid-rescaled variants depend on ``scale_dyn``
        (outside the hot path) and raise ``NotImplementedError``.
        """
        if lamsteps or velocity or trap:
            raise NotImplementedError("lamsteps / velocity / trap need Dynspec.scale_dyn, "
                                      "which is outside the accelerated hot path")
        if plot:
            raise NotImplementedError("plotting is outside the accelerated hot path")
        dyn = self.dyn if input_dyn is None else input_dyn
        dyn_t = to_device(dyn, torch.float64)
        sec = sspec_device(dyn_t, prewhite=prewhite, halve=halve, window=window,
                           window_frac=window_frac).cpu().numpy()
        nf, nt = dyn_t.shape
        nrfft = int(2 ** (np.ceil(np.log2(nf)) + 1))
        ncfft = int(2 ** (np.ceil(np.log2(nt)) + 1))
        td = np.array(list(range(0, int(nrfft / 2) if halve else int(nrfft))))
        fd = np.array(list(range(int(-ncfft / 2), int(ncfft / 2))))
        fdop = np.reshape(np.multiply(fd, 1e3 / (ncfft * self.dt)), [len(fd)])   # mHz
        tdel = np.reshape(np.divide(td, (nrfft * self.df)), [len(td)])           # us
        if input_dyn is None and not return_sspec:
            self.sspec = sec
            self.fdop = fdop
            self.tdel = tdel
            return None
        return fdop, tdel, sec
