"""Test-tooling stand-in for astropy (see units.py). NOT part of the product."""
from . import units  # noqa: F401
__version__ = "0.0-shim"
