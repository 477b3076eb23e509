"""Binary acceptance-angle apodization (mirror of bf/apod_methods/maxangle.py:17-39): 1 where the
angle between the element normal and the element->focus ray is <= max_angle (inclusive)."""
from __future__ import annotations

from dataclasses import dataclass

from ... import _native as nat
from ...engine import get_engine
from ...util import validate as v
from .apodmethod import ApodizationMethod, angle_kind


@dataclass
class MaxAngle(ApodizationMethod):
    max_angle: float = 30.0
    units: str = "deg"

    def __post_init__(self):
        v.number("Max angle", self.max_angle)
        v.non_negative("Max angle", self.max_angle)
        v.unit_kind(self.units, "angle", f"Units must be an angle type, got {self.units}.")

    def kernel_args(self):
        return angle_kind(nat.APOD_MAXANGLE, self.units), float(self.max_angle), 0.0

    def calc_apodization(self, arr, target, params=None, transform=None):
        _, apod = get_engine().beamform(arr, target, 1.0, transform=transform, apod=self.kernel_args())
        return apod if isinstance(target, (list, tuple)) else apod[0]
