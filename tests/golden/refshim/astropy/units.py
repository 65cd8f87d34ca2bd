"""Minimal stand-in for ``astropy.units`` -- TEST TOOLING ONLY.

astropy is not installable in the build container (no network), and every
module of the reference imports it at top level.  This file implements just
enough of the Quantity/Unit behaviour that the *unmodified* reference modules
under /root/reference can be imported and run by ``tests/golden/make_golden.py``
to generate golden vectors.  It is never imported by the product
(``scintools_amd``), by ``bench.py`` or by the GPU tests.

Numerical contract (what matters for the golden vectors): values are plain
float64; arithmetic between quantities is the plain NumPy ufunc on the
values; when two operands carry different-but-equivalent units the SECOND
operand is rescaled to the unit of the first by ``scale2/scale1`` and the
multiplication is skipped when that factor is exactly 1.0.  On the reference
hot path all such factors are 1.0 (us <-> s**3 mHz**2) or 1000.0 (1/s -> mHz).
"""
import numpy as np
from fractions import Fraction


class UnitConversionError(ValueError):
    pass


class UnitsError(ValueError):
    pass


class UnitBase:
    __array_ufunc__ = None          # make ndarray binops defer to __rmul__ etc.
    __array_priority__ = 100000

    def __init__(self, scale, powers, name=None):
        self.scale = float(scale)
        self.powers = {k: Fraction(v) for k, v in powers.items() if v != 0}
        self._name = name

    # -- algebra -----------------------------------------------------------
    def _combine(self, other, sign):
        p = dict(self.powers)
        for k, v in other.powers.items():
            p[k] = p.get(k, 0) + sign * v
        if sign > 0:
            sc = self.scale * other.scale
        else:
            sc = self.scale / other.scale
        return UnitBase(sc, p)

    def __mul__(self, other):
        if isinstance(other, UnitBase):
            return self._combine(other, +1)
        return Quantity(other, self)

    def __rmul__(self, other):
        if isinstance(other, Quantity):
            return Quantity(other.view(np.ndarray), other.unit * self)
        return Quantity(other, self)

    def __truediv__(self, other):
        if isinstance(other, UnitBase):
            return self._combine(other, -1)
        return Quantity(1.0 / np.asarray(other), self)

    def __rtruediv__(self, other):
        inv = self ** -1
        if isinstance(other, Quantity):
            return Quantity(other.view(np.ndarray), other.unit * inv)
        return Quantity(other, inv)

    def __pow__(self, p):
        p = Fraction(p).limit_denominator(1000)
        return UnitBase(self.scale ** float(p),
                        {k: v * p for k, v in self.powers.items()})

    # -- comparisons -------------------------------------------------------
    def is_equivalent(self, other):
        return self.powers == other.powers

    def _to(self, other):
        if not self.is_equivalent(other):
            raise UnitConversionError(f"'{self}' and '{other}' are not convertible")
        return self.scale / other.scale

    def to(self, other, value=1.0):
        return value * self._to(other)

    def __eq__(self, other):
        return (isinstance(other, UnitBase) and self.powers == other.powers
                and self.scale == other.scale)

    def __hash__(self):
        return hash((self.scale, tuple(sorted(self.powers.items()))))

    def __repr__(self):
        if self._name:
            return self._name
        body = " ".join(f"{k}{'' if v == 1 else v}" for k, v in sorted(self.powers.items()))
        return f"{self.scale:g} {body}".strip()

    __str__ = __repr__


dimensionless_unscaled = UnitBase(1.0, {}, "")
one = dimensionless_unscaled
s = UnitBase(1.0, {"s": 1}, "s")
ms = UnitBase(1e-3, {"s": 1}, "ms")
us = UnitBase(1e-6, {"s": 1}, "us")
ns = UnitBase(1e-9, {"s": 1}, "ns")
minute = UnitBase(60.0, {"s": 1}, "min")
hour = UnitBase(3600.0, {"s": 1}, "h")
day = UnitBase(86400.0, {"s": 1}, "d")
yr = UnitBase(86400.0 * 365.25, {"s": 1}, "yr")
Hz = UnitBase(1.0, {"s": -1}, "Hz")
mHz = UnitBase(1e-3, {"s": -1}, "mHz")
kHz = UnitBase(1e3, {"s": -1}, "kHz")
MHz = UnitBase(1e6, {"s": -1}, "MHz")
GHz = UnitBase(1e9, {"s": -1}, "GHz")
m = UnitBase(1.0, {"m": 1}, "m")
km = UnitBase(1e3, {"m": 1}, "km")
rad = UnitBase(1.0, {}, "rad")
deg = UnitBase(np.pi / 180.0, {}, "deg")
mas = UnitBase(np.pi / 180.0 / 3600e3, {}, "mas")
kpc = UnitBase(3.0856775814913674e19, {"m": 1}, "kpc")
pc = UnitBase(3.0856775814913674e16, {"m": 1}, "pc")
Unit = UnitBase

_SAME_UNIT = {"add", "subtract", "minimum", "maximum", "fmin", "fmax",
              "remainder", "fmod", "hypot"}
_COMPARE = {"less", "greater", "less_equal", "greater_equal", "equal", "not_equal"}
_KEEP = {"absolute", "fabs", "negative", "positive", "conjugate", "conj", "floor",
         "ceil", "rint", "trunc", "real", "imag", "copysign"}
_PLAIN = {"isfinite", "isnan", "isinf", "sign", "signbit"}
_DIMLESS_IN = {"sin", "cos", "tan", "exp", "log", "log10", "log2", "arcsin",
               "arccos", "arctan", "exp2", "expm1", "log1p", "sinh", "cosh", "tanh"}


def _unit_of(x):
    return x.unit if isinstance(x, Quantity) else dimensionless_unscaled


def _val(x):
    return x.view(np.ndarray) if isinstance(x, Quantity) else x


class Quantity(np.ndarray):
    __array_priority__ = 10000

    def __new__(cls, value, unit=None, dtype=None, copy=True):
        if isinstance(value, Quantity):
            if unit is None:
                unit = value.unit
                value = value.view(np.ndarray)
            else:
                value = value.to_value(unit)
        arr = np.array(value, dtype=dtype, copy=True, subok=False)
        if arr.dtype.kind in "iub" and dtype is None:
            arr = arr.astype(float)
        obj = arr.view(cls)
        obj._unit = unit if unit is not None else dimensionless_unscaled
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self._unit = getattr(obj, "_unit", dimensionless_unscaled)

    # -- basic accessors -----------------------------------------------------
    @property
    def unit(self):
        return self._unit

    @property
    def value(self):
        v = self.view(np.ndarray)
        return v[()] if v.ndim == 0 else v

    def _wrap(self, v, unit=None):
        out = np.asarray(v).view(Quantity)
        out._unit = self._unit if unit is None else unit
        return out

    def to(self, unit):
        if isinstance(unit, Quantity):      # astropy accepts `1 / (u.m * u.mHz**2)` as a unit
            unit = UnitBase(unit.unit.scale * float(unit.value), unit.unit.powers)
        f = self._unit._to(unit)
        v = self.view(np.ndarray)
        v = v.copy() if f == 1.0 else v * f
        return self._wrap(v, unit)

    def to_value(self, unit=None):
        if unit is None:
            return self.value
        return self.to(unit).value

    def decompose(self):
        base = UnitBase(1.0, self._unit.powers)
        return self.to(base)

    @property
    def si(self):
        return self.decompose()

    # -- python protocol -------------------------------------------------------
    def __getitem__(self, key):
        out = super().__getitem__(key)
        if not isinstance(out, Quantity):
            out = self._wrap(out)
        return out

    def __setitem__(self, key, value):
        if isinstance(value, Quantity):
            value = value.to_value(self._unit)
        self.view(np.ndarray)[key] = value

    def __iter__(self):
        if self.ndim == 0:
            raise TypeError("scalar quantity is not iterable")
        for i in range(self.shape[0]):
            yield self[i]

    def __float__(self):
        return float(self.to_value(dimensionless_unscaled))

    def __int__(self):
        return int(self.to_value(dimensionless_unscaled))

    def __index__(self):
        return int(self.view(np.ndarray))

    def __bool__(self):
        return bool(self.view(np.ndarray))

    def __repr__(self):
        return f"<Quantity {self.view(np.ndarray)!r} {self._unit}>"

    def __str__(self):
        return f"{self.view(np.ndarray)} {self._unit}"

    def __format__(self, spec):
        return f"{format(self.value, spec) if self.ndim == 0 else self.value} {self._unit}"

    def __mul__(self, other):
        if isinstance(other, UnitBase):
            return self._wrap(self.view(np.ndarray).copy(), self._unit * other)
        return super().__mul__(other)

    __rmul__ = __mul__

    def __imul__(self, other):
        if isinstance(other, UnitBase):
            self._unit = self._unit * other
            return self
        return super().__imul__(other)

    def __truediv__(self, other):
        if isinstance(other, UnitBase):
            return self._wrap(self.view(np.ndarray).copy(), self._unit / other)
        return super().__truediv__(other)

    def __itruediv__(self, other):
        if isinstance(other, UnitBase):
            self._unit = self._unit / other
            return self
        return super().__itruediv__(other)

    # -- reductions (keep the unit) ---------------------------------------------
    def _reduce(self, fn, *a, **k):
        return self._wrap(fn(self.view(np.ndarray), *a, **k))

    def mean(self, *a, **k):
        return self._reduce(np.mean, *a, **k)

    def max(self, *a, **k):
        return self._reduce(np.max, *a, **k)

    def min(self, *a, **k):
        return self._reduce(np.min, *a, **k)

    def sum(self, *a, **k):
        return self._reduce(np.sum, *a, **k)

    def std(self, *a, **k):
        return self._reduce(np.std, *a, **k)

    def ptp(self, *a, **k):
        return self._reduce(np.ptp, *a, **k)

    # -- ufunc dispatch ---------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        name = ufunc.__name__
        out = kwargs.pop("out", None)
        units = [_unit_of(x) for x in inputs]
        vals = [_val(x) for x in inputs]
        if out is not None:
            kwargs["out"] = tuple(_val(o) for o in out)

        def conv_second_to_first():
            u0, u1 = units
            # astropy lets a bare 0 / inf / nan combine with any unit
            if not isinstance(inputs[1], Quantity) and not u0.is_equivalent(u1):
                a = np.asarray(inputs[1])
                if np.all((a == 0) | ~np.isfinite(a)):
                    return u0
                raise UnitConversionError(
                    f"Can only apply '{name}' function to dimensionless quantities "
                    f"when other argument is not a quantity")
            if not isinstance(inputs[0], Quantity) and not u0.is_equivalent(u1):
                a = np.asarray(inputs[0])
                if np.all((a == 0) | ~np.isfinite(a)):
                    return u1
                raise UnitConversionError(f"incompatible units in '{name}'")
            f = u1._to(u0)
            if f != 1.0:
                vals[1] = vals[1] * f
            return u0

        if method == "reduce":
            res = ufunc.reduce(*vals, **kwargs)
            if name in ("add", "maximum", "minimum", "fmax", "fmin"):
                unit = units[0]
            elif name in ("logical_and", "logical_or"):
                return res
            else:
                raise NotImplementedError(f"shim: reduce of {name}")
        elif method == "accumulate" and name == "add":
            res = ufunc.accumulate(*vals, **kwargs)
            unit = units[0]
        elif method != "__call__":
            raise NotImplementedError(f"shim: ufunc method {method} for {name}")
        elif name == "multiply":
            res = ufunc(*vals, **kwargs)
            unit = units[0] * units[1]
        elif name in ("true_divide", "divide"):
            res = ufunc(*vals, **kwargs)
            unit = units[0] / units[1]
        elif name == "floor_divide":
            conv_second_to_first()
            res = ufunc(*vals, **kwargs)
            unit = dimensionless_unscaled
        elif name in _SAME_UNIT:
            unit = conv_second_to_first()
            res = ufunc(*vals, **kwargs)
        elif name in _COMPARE:
            conv_second_to_first()
            return ufunc(*vals, **kwargs)
        elif name == "power":
            p = vals[1]
            if units[1].powers:
                raise UnitsError("exponent must be dimensionless")
            res = ufunc(*vals, **kwargs)
            unit = units[0] ** float(np.asarray(p).ravel()[0]) if units[0].powers or units[0].scale != 1 else units[0]
        elif name == "square":
            res = ufunc(*vals, **kwargs)
            unit = units[0] ** 2
        elif name == "sqrt":
            res = ufunc(*vals, **kwargs)
            unit = units[0] ** Fraction(1, 2)
        elif name == "reciprocal":
            res = ufunc(*vals, **kwargs)
            unit = units[0] ** -1
        elif name in _KEEP:
            res = ufunc(*vals, **kwargs)
            unit = units[0]
        elif name in _PLAIN:
            return ufunc(*vals, **kwargs)
        elif name in _DIMLESS_IN:
            f = units[0]._to(dimensionless_unscaled)
            res = ufunc(vals[0] * f if f != 1.0 else vals[0], **kwargs)
            unit = dimensionless_unscaled
        else:
            raise NotImplementedError(f"shim: ufunc {name} not supported on Quantity")

        if out is not None:
            o = out[0]
            if isinstance(o, Quantity):
                o._unit = unit
            return o
        return self._wrap(res, unit)
