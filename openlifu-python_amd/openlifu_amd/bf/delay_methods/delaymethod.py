"""Base of the delay-method family (mirror of bf/delay_methods/delaymethod.py:15-41)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass

from ...util.plugin import ClassTagged, lookup


@dataclass
class DelayMethod(ClassTagged, ABC):
    @abstractmethod
    def calc_delays(self, arr, target, params, transform=None):
        """delays[N] in seconds for one focus."""

    @staticmethod
    def from_dict(d):
        cls, kwargs = lookup(__package__, d)
        return cls(**kwargs)
