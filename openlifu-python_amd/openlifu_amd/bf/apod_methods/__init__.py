"""Apodization plug-ins (class-name lookup namespace for ``ApodizationMethod.from_dict``).

``Uniform`` needs no geometry; ``MaxAngle`` and ``PiecewiseLinear`` get the element-normal-to-ray angle
from HIP kernel 1.
"""
from __future__ import annotations

from . import apodmethod as _base
from . import maxangle as _maxangle
from . import piecewiselinear as _pwl
from . import uniform as _uniform

ApodizationMethod = _base.ApodizationMethod
Uniform = _uniform.Uniform
MaxAngle = _maxangle.MaxAngle
PiecewiseLinear = _pwl.PiecewiseLinear

__all__ = ("ApodizationMethod", "Uniform", "MaxAngle", "PiecewiseLinear")
