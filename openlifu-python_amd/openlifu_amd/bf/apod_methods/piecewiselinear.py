"""Piecewise-linear roll-off apodization (mirror of bf/apod_methods/piecewiselinear.py:17-49):
clip((zero_angle - theta) / (zero_angle - rolloff_angle), 0, 1)."""
from __future__ import annotations

from dataclasses import dataclass

from ... import _native as nat
from ...engine import get_engine
from ...util import validate as v
from .apodmethod import ApodizationMethod, angle_kind


@dataclass
class PiecewiseLinear(ApodizationMethod):
    zero_angle: float = 90.0     # at and beyond this angle the weight is 0
    rolloff_angle: float = 45.0  # below this angle the weight is 1
    units: str = "deg"

    def __post_init__(self):
        for label, value in (("Zero angle", self.zero_angle), ("Rolloff angle", self.rolloff_angle)):
            v.number(label, value)
            v.non_negative(label, value)
        if self.rolloff_angle >= self.zero_angle:
            raise ValueError(f"Rolloff angle must be less than zero angle, got {self.rolloff_angle} >= {self.zero_angle}.")
        v.unit_kind(self.units, "angle", f"Units must be an angle type, got {self.units}.")

    def kernel_args(self):
        return angle_kind(nat.APOD_PIECEWISE, self.units), float(self.zero_angle), float(self.rolloff_angle)

    def calc_apodization(self, arr, target, params=None, transform=None):
        _, apod = get_engine().beamform(arr, target, 1.0, transform=transform, apod=self.kernel_args())
        return apod if isinstance(target, (list, tuple)) else apod[0]
