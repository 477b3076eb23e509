"""Binary acceptance-angle apodization (mirror of bf/apod_methods/maxangle.py:17-39): 1 where the
angle between the element normal and the element->focus ray is <= max_angle (inclusive)."""
from __future__ import annotations

from dataclasses import dataclass

from ... import _native as nat
from ...engine import get_engine
from ...util.units import getunittype
from .apodmethod import ApodizationMethod, angle_kind


@dataclass
class MaxAngle(ApodizationMethod):
    max_angle: float = 30.0
    units: str = "deg"

    def __post_init__(self):
        if not isinstance(self.max_angle, (int, float)):
            raise TypeError(f"Max angle must be a number, got {type(self.max_angle).__name__}.")
        if self.max_angle < 0:
            raise ValueError(f"Max angle must be non-negative, got {self.max_angle}.")
        if getunittype(self.units) != "angle":
            raise ValueError(f"Units must be an angle type, got {self.units}.")

    def kernel_args(self):
        return angle_kind(nat.APOD_MAXANGLE, self.units), float(self.max_angle), 0.0

    def calc_apodization(self, arr, target, params=None, transform=None):
        _, apod = get_engine().beamform(arr, target, 1.0, transform=transform, apod=self.kernel_args())
        return apod if isinstance(target, (list, tuple)) else apod[0]
