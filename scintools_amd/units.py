"""Unit handling at the boundary (ththmod.unit_checks, ththmod.py:1639-1668).

With astropy installed the behaviour is the reference's: bare numbers are
assumed to be in the desired unit (with a warning), equivalent units are
converted, incompatible units raise ``astropy.units.UnitConversionError``.
Without astropy only bare numbers are accepted.  Either way the kernels get
plain float64 in us / mHz / s**3 / MHz / s.
"""
import warnings

import numpy as np

try:  # pragma: no cover - astropy is absent from the build container
    import astropy.units as u
    HAVE_ASTROPY = True
except Exception:  # pragma: no cover
    u = None
    HAVE_ASTROPY = False

_NAMES = {"us": "us", "mHz": "mHz", "s3": "s3", "MHz": "MHz", "s": "s"}


def unit_of(key):
    """astropy unit for one of the keys 'us', 'mHz', 's3', 'MHz', 's' (None without astropy)."""
    if not HAVE_ASTROPY:
        return None
    return {"us": u.us, "mHz": u.mHz, "s3": u.s**3, "MHz": u.MHz, "s": u.s}[key]


def strip(var, name, key, warn=True):
    """Return float64 ndarray (or 0-d array) of `var` expressed in unit `key`."""
    if HAVE_ASTROPY and hasattr(var, "unit"):
        desired = unit_of(key)
        if u.dimensionless_unscaled.is_equivalent(var.unit):
            if warn:
                warnings.warn(f"{name} missing units. Assuming {desired}.")
            return np.asarray(var.value, dtype=float)
        if desired.is_equivalent(var.unit):
            return np.asarray(var.to(desired).value, dtype=float)
        raise u.UnitConversionError(f"{name} units ({var.unit}) not equivalent to {desired}")
    if hasattr(var, "unit") and hasattr(var, "value"):
        raise TypeError(f"{name} carries units but astropy is not importable")
    if warn and HAVE_ASTROPY:
        warnings.warn(f"{name} missing units. Assuming {_NAMES[key]}.")
    return np.asarray(var, dtype=float)


def attach(val, key):
    """Give a result its reference unit when astropy is present."""
    if HAVE_ASTROPY:
        return val * unit_of(key)
    return val
