"""CPU oracle for the secondary-spectrum half of the hot path -- TEST INFRASTRUCTURE.

Plain NumPy/SciPy restatement of ``Dynspec.calc_sspec`` (dynspec.py:3665-3721),
``scint_utils.get_window`` (scint_utils.py:810-832) and, for the CPU-only
"plumbing" config, ``Dynspec.calc_acf`` (dynspec.py:3780-3797).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

PARITY PIN: checked against ``tests/golden/sspec_*.npz`` -- outputs of the
unmodified reference ``Dynspec.calc_sspec`` / ``calc_acf`` run in the build
container on a seeded reference ``Simulation`` (tests/golden/make_golden.py).
"""
import numpy as np
import scipy.constants as sc
from scipy.signal import convolve2d

_WINDOWS = {"hanning": np.hanning, "hamming": np.hamming,
            "blackman": np.blackman, "bartlett": np.bartlett}


def get_window(nt, nf, window="hanning", frac=0.1):
    """Edge tapers with a flat middle (scint_utils.py:810-832).
    Returns (chan_window[nt], subint_window[nf])."""
    fn = _WINDOWS[window.lower()]
    cw = fn(np.floor(frac * nt))
    sw = fn(np.floor(frac * nf))
    chan_window = np.insert(cw, int(np.ceil(len(cw) / 2)), np.ones([nt - len(cw)]))
    subint_window = np.insert(sw, int(np.ceil(len(sw) / 2)), np.ones([nf - len(sw)]))
    return chan_window, subint_window


def fft_lengths(nf, nt):
    """Padded FFT sizes (dynspec.py:3677-3678): 2 * next power of two."""
    nrfft = int(2 ** (np.ceil(np.log2(nf)) + 1))
    ncfft = int(2 ** (np.ceil(np.log2(nt)) + 1))
    return nrfft, ncfft


def calc_sspec(dyn, dt, df, prewhite=False, halve=True, window="hanning", window_frac=0.1):
    """Secondary spectrum in dB (dynspec.py:3665-3721).

    dyn[nf, nt] float64, dt [s], df [MHz].  Returns (fdop [mHz], tdel [us], sec).
    """
    nf, nt = np.shape(dyn)
    dyn = dyn - np.mean(dyn)
    if window is not None:
        chan_window, subint_window = get_window(nt, nf, window=window, frac=window_frac)
        dyn = np.multiply(chan_window, dyn)
        dyn = np.transpose(np.multiply(subint_window, np.transpose(dyn)))
    nrfft, ncfft = fft_lengths(nf, nt)
    dyn = dyn - np.mean(dyn)
    if prewhite:
        simpw = convolve2d([[1, -1], [-1, 1]], dyn, mode="valid")
    else:
        simpw = dyn
    simf = np.fft.fft2(simpw, s=[nrfft, ncfft])
    simf = np.real(np.multiply(simf, np.conj(simf)))
    sec = np.fft.fftshift(simf)
    if halve:
        sec = sec[int(nrfft / 2):][:]
        td = np.arange(0, int(nrfft / 2))
    else:
        td = np.arange(0, int(nrfft))
    fd = np.arange(int(-ncfft / 2), int(ncfft / 2))
    fdop = np.reshape(np.multiply(fd, 1e3 / (ncfft * dt)), [len(fd)])
    tdel = np.reshape(np.divide(td, (nrfft * df)), [len(td)])
    if prewhite:
        if not halve:
            raise RuntimeError("Cannot apply prewhite to full frame")
        vec1 = np.reshape(np.power(np.sin(np.multiply(sc.pi / ncfft, fd)), 2), [ncfft, 1])
        vec2 = np.reshape(np.power(np.sin(np.multiply(sc.pi / nrfft, td)), 2),
                          [1, int(nrfft / 2)])
        postdark = np.transpose(vec1 * vec2)
        postdark[:, int(ncfft / 2)] = 1
        postdark[0, :] = 1
        sec = np.divide(sec, postdark)
    with np.errstate(divide="ignore"):
        sec = 10 * np.log10(sec)
    return fdop, tdel, sec


def calc_acf(dyn, normalise=True):
    """Direct-method autocovariance (dynspec.py:3780-3797); CPU-only plumbing."""
    valid = np.isfinite(dyn)
    arr = dyn - np.mean(dyn[valid])
    nf, nt = dyn.shape
    arr = np.fft.fft2(arr, s=[2 * nf, 2 * nt])
    arr = np.abs(arr)
    arr **= 2
    arr = np.fft.ifft2(arr)
    arr = np.fft.fftshift(arr)
    arr = np.real(arr)
    if normalise:
        arr /= np.max(arr)
    return arr
