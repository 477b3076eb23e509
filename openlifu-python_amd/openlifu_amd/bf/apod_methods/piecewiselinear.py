"""Piecewise-linear roll-off apodization (mirror of bf/apod_methods/piecewiselinear.py:17-49):
clip((zero_angle - theta) / (zero_angle - rolloff_angle), 0, 1)."""
from __future__ import annotations

from dataclasses import dataclass

from ... import _native as nat
from ...engine import get_engine
from ...util.units import getunittype
from .apodmethod import ApodizationMethod, angle_kind


@dataclass
class PiecewiseLinear(ApodizationMethod):
    zero_angle: float = 90.0
    rolloff_angle: float = 45.0
    units: str = "deg"

    def __post_init__(self):
        if not isinstance(self.zero_angle, (int, float)):
            raise TypeError(f"Zero angle must be a number, got {type(self.zero_angle).__name__}.")
        if self.zero_angle < 0:
            raise ValueError(f"Zero angle must be non-negative, got {self.zero_angle}.")
        if not isinstance(self.rolloff_angle, (int, float)):
            raise TypeError(f"Rolloff angle must be a number, got {type(self.rolloff_angle).__name__}.")
        if self.rolloff_angle < 0:
            raise ValueError(f"Rolloff angle must be non-negative, got {self.rolloff_angle}.")
        if self.rolloff_angle >= self.zero_angle:
            raise ValueError(f"Rolloff angle must be less than zero angle, got {self.rolloff_angle} >= {self.zero_angle}.")
        if getunittype(self.units) != "angle":
            raise ValueError(f"Units must be an angle type, got {self.units}.")

    def kernel_args(self):
        return angle_kind(nat.APOD_PIECEWISE, self.units), float(self.zero_angle), float(self.rolloff_angle)

    def calc_apodization(self, arr, target, params=None, transform=None):
        _, apod = get_engine().beamform(arr, target, 1.0, transform=transform, apod=self.kernel_args())
        return apod if isinstance(target, (list, tuple)) else apod[0]
