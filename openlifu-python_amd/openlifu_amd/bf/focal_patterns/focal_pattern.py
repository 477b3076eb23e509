"""Base of the focal-pattern family (mirror of bf/focal_patterns/focal_pattern.py:15-85):
``target_pressure`` (+ units) is what ``Solution.scale`` steers the mainlobe peak to."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass

from ...util.plugin import ClassTagged, lookup
from ...util.validate import positive, unit_kind


@dataclass
class FocalPattern(ClassTagged, ABC):
    target_pressure: float = 1.0
    units: str = "Pa"

    def __post_init__(self):
        positive("Target pressure must be greater than 0", self.target_pressure)
        if not isinstance(self.units, str):
            raise TypeError("Units must be a string")
        unit_kind(self.units, "pressure", f"Units must be a pressure unit, got {self.units}")

    @abstractmethod
    def get_targets(self, target):
        """list of foci (Points) for one target."""

    @abstractmethod
    def num_foci(self):
        """number of foci the pattern expands to."""

    @staticmethod
    def from_dict(d):
        cls, kwargs = lookup(__package__, d)
        return cls(**kwargs)
