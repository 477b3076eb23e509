"""Delay-method plug-in base (mirror of bf/delay_methods/delaymethod.py:15-41): subclasses are
found by class name in this package's namespace (``{"class": "Direct", ...}``)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass


@dataclass
class DelayMethod(ABC):
    @abstractmethod
    def calc_delays(self, arr, target, params, transform=None):
        ...

    def to_dict(self):
        d = self.__dict__.copy()
        d["class"] = self.__class__.__name__
        return d

    @staticmethod
    def from_dict(d):
        from .. import delay_methods
        d = d.copy()
        return getattr(delay_methods, d.pop("class"))(**d)
