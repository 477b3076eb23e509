"""Base of the apodization family (mirror of bf/apod_methods/apodmethod.py:16-42)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass

from ...util.plugin import ClassTagged, lookup
from ...util.units import getunittype


@dataclass
class ApodizationMethod(ClassTagged, ABC):
    @abstractmethod
    def calc_apodization(self, arr, target, params, transform=None):
        """weights[N] in [0, 1] for one focus."""

    @abstractmethod
    def kernel_args(self):
        """(apod_kind, p0, p1) for olx_bf_solve (include/olx.h)."""

    @staticmethod
    def from_dict(d):
        cls, kwargs = lookup(__package__, d)
        return cls(**kwargs)


def angle_kind(base_kind: int, units: str) -> int:
    """deg is the kernel's default; any radian spelling sets OLX_APOD_RADIANS."""
    if getunittype(units) != "angle":
        raise ValueError(f"Units must be an angle type, got {units}.")
    return base_kind | (0x10 if units.lower().startswith("rad") else 0)
