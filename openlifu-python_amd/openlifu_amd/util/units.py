"""Unit-string parsing and scale factors.

Mirror of the reference's ``openlifu.util.units`` (util/units.py:7-179) -- same
function names, argument meaning and error behaviour, including its quirks (e.g.
``"micron"`` classifies as a distance but cannot be scaled, units.py:101-107).
Golden G6 (tests/golden/g6_units.json, produced by the real reference) pins the
scale factors bit for bit: mm->m must be exactly 1e-3 / 1.0.
"""
from __future__ import annotations

import numpy as np

_TIME_WORDS = {"minute", "minutes", "min", "mins", "hour", "hours", "hr", "hrs", "day", "days", "d"}
_ANGLE_WORDS = {"rad", "deg", "radian", "radians", "degree", "degrees", "°"}

# prefix -> SI scale (util/units.py:137-169)
_PREFIX = {
    "pico": 1e-12, "p": 1e-12, "nano": 1e-9, "n": 1e-9,
    "micro": 1e-6, "u": 1e-6, "µ": 1e-6, "μ": 1e-6,
    "milli": 1e-3, "m": 1e-3, "centi": 1e-2, "c": 1e-2,
    "kilo": 1e3, "k": 1e3, "mega": 1e6, "M": 1e6, "giga": 1e9, "G": 1e9, "tera": 1e12, "T": 1e12,
    "min": 60.0, "minute": 60.0, "hour": 3600.0, "hr": 3600.0, "day": 86400.0, "d": 86400.0,
    "rad": 1.0, "radian": 1.0, "radians": 1.0,
    "deg": 2 * 3.14159265358979323846 / 360, "degree": 2 * 3.14159265358979323846 / 360,
    "degrees": 2 * 3.14159265358979323846 / 360, "°": 2 * 3.14159265358979323846 / 360,
}


def getunittype(unit: str) -> str:
    """Classify a unit string (util/units.py:7-34; rule order matters)."""
    u = unit.lower()
    if u in ("micron", "microns"):
        return "distance"
    if u in _TIME_WORDS:
        return "time"
    if u in _ANGLE_WORDS:
        return "angle"
    if "sec" in u:
        return "time"
    if "meter" in u or "micron" in u:
        return "distance"
    for suffix, kind in ((("s",), "time"), (("m",), "distance"), (("m2", "m^2"), "area"),
                         (("m3", "m^3"), "volume"), (("hz",), "frequency"), (("pa",), "pressure"),
                         (("w",), "watt")):
        if u.endswith(suffix):
            return kind
    return "other"


def _first_found(unit: str, needles, reverse_last=None):
    for n in needles:
        i = unit.find(n)
        if i != -1:
            return i
    if reverse_last is not None:
        i = unit.rfind(reverse_last)
        if i != -1:
            return i
    return len(unit)


def getsiscale(unit: str, type: str) -> float:  # noqa: A002 - reference argument name
    """SI scale of ``unit`` given its type (util/units.py:96-179)."""
    kind = type.lower()
    if kind in ("distance", "area", "volume"):
        idx = _first_found(unit, ("meters", "meter"))
        if idx == len(unit) and "meter" not in unit:
            idx = 6 if unit.lower() == "micron" else _first_found(unit, (), reverse_last="m")
    elif kind == "time":
        idx = _first_found(unit, ("seconds", "second", "sec"), reverse_last="s")
    elif kind == "angle":
        idx = len(unit)
    elif kind in ("frequency", "pressure"):
        idx = len(unit) - 2
    elif kind == "watt":
        idx = len(unit) - 1
    else:
        idx = len(unit) - len(kind) + 1
    prefix = unit[:idx]
    if not prefix:
        scl = 1.0
    elif prefix in _PREFIX:
        scl = _PREFIX[prefix]
    else:
        raise ValueError(f"Unknown prefix {prefix}")
    if kind == "area":
        scl = scl ** 2.0
    elif kind == "volume":
        scl = scl ** 3.0
    return scl


_CONV_CACHE = {}


def getunitconversion(from_unit, to_unit, unitratio=None, constant=None) -> float:
    """Multiplicative factor from ``from_unit`` to ``to_unit`` (util/units.py:36-94).  Plain string pairs are memoised:
    the host mirror asks for the same few conversions once per element and call (a 256-element calc_solution made ~1000
    of them, 5 ms); errors are not cached."""
    if not from_unit:
        return 1.0
    if unitratio is None and constant is None and isinstance(from_unit, str) and isinstance(to_unit, str):
        key = (from_unit, to_unit)
        hit = _CONV_CACHE.get(key)
        if hit is None:
            hit = _CONV_CACHE[key] = _getunitconversion(from_unit, to_unit)
        return hit
    return _getunitconversion(from_unit, to_unit, unitratio, constant)


def _getunitconversion(from_unit, to_unit, unitratio=None, constant=None) -> float:
    if unitratio is not None and constant is not None:
        if "/" not in unitratio:
            raise ValueError("Conversion unit ratio must have a '/' symbol")
        unitn, unitd = unitratio.split("/")
        t0, t1, tn, td = (getunittype(u) for u in (from_unit, to_unit, unitn, unitd))
        if t0 == td and t1 == tn:
            return getunitconversion(from_unit, unitd) * constant * getunitconversion(unitn, to_unit)
        if t0 == tn and t1 == td:
            return getunitconversion(from_unit, unitn) * 1 / constant * getunitconversion(unitd, to_unit)
        if t0 == t1:
            return getunitconversion(from_unit, to_unit)
        raise ValueError(f"Unit type mismatch {t0} -> ({tn}/{td}) -> {t1}")
    s0, s1 = from_unit.find("/"), to_unit.find("/")
    if s0 != -1 and s1 != -1:
        return (getunitconversion(from_unit[:s0], to_unit[:s1])
                / getunitconversion(from_unit[s0 + 1:], to_unit[s1 + 1:]))
    if s0 != -1 or s1 != -1:
        raise ValueError(f"Unit ratio mismatch ({from_unit} vs {to_unit})")
    t0, t1 = getunittype(from_unit), getunittype(to_unit)
    if t0 != t1:
        raise ValueError(f"Unit type mismatch ({t0}) vs ({t1})")
    if t0 != "other":
        return getsiscale(from_unit, t0) / getsiscale(to_unit, t0)
    if from_unit[-1] != to_unit[-1]:
        raise ValueError(f"Cannot convert {from_unit} to {to_unit}")
    # longest common suffix names the base unit (units.py:76-79)
    i = 0
    base = from_unit
    while i < min(len(from_unit), len(to_unit)) and from_unit[-i:] == to_unit[-i:]:
        base = from_unit[-i:]
        i += 1
    return getsiscale(from_unit, base) / getsiscale(to_unit, base)


def rescale_data_arr(data_arr, units: str):
    """Copy of ``data_arr`` with its values converted to ``units`` (util/units.py:182-198): ``attrs['units']`` names the current units."""
    rescaled = data_arr.copy(deep=True)
    scale = getunitconversion(data_arr.attrs["units"], units)
    rescaled.data *= scale
    rescaled.attrs["units"] = units
    return rescaled


def rescale_coords(data_arr, units: str):
    """Copy of ``data_arr`` with every coordinate that carries ``attrs['units']`` converted to ``units`` (util/units.py:200-222)."""
    rescaled = data_arr.copy(deep=True)
    for key in list(data_arr.coords):
        attrs = rescaled.coords[key].attrs
        if "units" in attrs:
            scale = getunitconversion(attrs["units"], units)
            new_attrs = dict(attrs)
            new_attrs["units"] = units
            values = scale * np.asarray(rescaled.coords[key].data)
            if hasattr(rescaled, "assign_coords") and type(rescaled).__module__.startswith("xarray"):     # pragma: no cover - xarray absent in the image
                rescaled = rescaled.assign_coords({key: (key, values, new_attrs)})
            else:
                c = rescaled.coords[key]
                c.data = values
                c.attrs = new_attrs
    return rescaled
