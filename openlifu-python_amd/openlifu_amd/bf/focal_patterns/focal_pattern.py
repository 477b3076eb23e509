"""Focal-pattern plug-in base (mirror of bf/focal_patterns/focal_pattern.py:15-85)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass

from ...util.units import getunittype


@dataclass
class FocalPattern(ABC):
    target_pressure: float = 1.0
    units: str = "Pa"

    def __post_init__(self):
        if self.target_pressure <= 0:
            raise ValueError("Target pressure must be greater than 0")
        if not isinstance(self.units, str):
            raise TypeError("Units must be a string")
        if getunittype(self.units) != "pressure":
            raise ValueError(f"Units must be a pressure unit, got {self.units}")

    @abstractmethod
    def get_targets(self, target):
        ...

    @abstractmethod
    def num_foci(self):
        ...

    def to_dict(self):
        d = self.__dict__.copy()
        d["class"] = self.__class__.__name__
        return d

    @staticmethod
    def from_dict(d):
        from .. import focal_patterns
        d = d.copy()
        return getattr(focal_patterns, d.pop("class"))(**d)
