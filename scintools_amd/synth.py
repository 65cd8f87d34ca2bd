"""Seeded synthetic dynamic spectra with a known arc curvature.

This is the "known-answer input" of SURVEY.md section 8(d): a 1-D screen of
``nimg`` images at Doppler shifts theta_k [mHz] and delays eta*theta_k**2 [us],

    E(f, t) = sum_k a_k exp(2 pi i [ (f - f0) eta theta_k**2 + t theta_k 1e-3 ]),
    dyn     = |E|**2 + noise,

so that the theta-theta eigenvalue curve lambda(eta) peaks at ``eta_true``.
It is host-side NumPy (two small matrix products), used by bench.py and the
tests to make inputs of any size without the reference's simulator.
"""
import numpy as np


def arc_axes(nf, nt, eta_true=0.02, df=None, dt=30.0, f0=1400.0, theta_max=None):
    """(freqs[nf] MHz, times[nt] s, theta_max mHz, df MHz) of :func:`arc_dynspec` -- the axes alone, for callers that
    need the grids of a workload (crop sizes N per curvature, sharding models) without synthesising its pixels."""
    fd_max = 1e3 / (2 * dt)
    if theta_max is None:
        theta_max = fd_max / 2
    if df is None:
        df = 0.7 / (2 * eta_true * theta_max**2)
    freqs = f0 + (np.arange(nf) - nf // 2) * df
    times = dt * np.arange(nt)
    return freqs, times, theta_max, df


def arc_dynspec(nf, nt, seed=0, eta_true=0.02, nimg=64, df=None, dt=30.0,
                f0=1400.0, theta_max=None, noise=1.0):
    """Return (dyn[nf, nt] float64, freqs[nf] MHz, times[nt] s, eta_true s**3).

    Defaults are sized so the arc stays inside the conjugate spectrum:
    fd_max = 1e3/(2 dt) mHz, theta_max = fd_max/2 and df chosen so that
    eta_true*theta_max**2 is 70 % of tau_max = 1/(2 df) us.
    """
    rng = np.random.default_rng(seed)
    freqs, times, theta_max, df = arc_axes(nf, nt, eta_true, df, dt, f0, theta_max)
    theta = rng.uniform(-theta_max, theta_max, nimg)
    amp = (rng.standard_normal(nimg) + 1j * rng.standard_normal(nimg))
    amp *= np.exp(-(theta / (theta_max / 2)) ** 2)
    theta[0] = 0.0
    amp[0] = 8.0
    U = np.exp(2j * np.pi * np.outer(freqs - f0, eta_true * theta**2))  # [nf, nimg]
    V = np.exp(2j * np.pi * np.outer(theta * 1e-3, times))              # [nimg, nt]
    E = (U * amp) @ V
    dyn = E.real**2 + E.imag**2
    if noise:
        dyn = dyn + noise * rng.standard_normal(dyn.shape)
    return np.ascontiguousarray(dyn), freqs, times, eta_true
