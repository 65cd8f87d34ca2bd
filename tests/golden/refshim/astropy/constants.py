"""Stand-in for astropy.constants: only what the reference touches at import/use."""
from .units import Quantity, m, s
c = Quantity(299792458.0, m / s)
