from __future__ import annotations

from .apodmethod import ApodizationMethod
from .maxangle import MaxAngle
from .piecewiselinear import PiecewiseLinear
from .uniform import Uniform

__all__ = ["ApodizationMethod", "MaxAngle", "PiecewiseLinear", "Uniform"]
