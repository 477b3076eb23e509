"""Uniform apodization (mirror of bf/apod_methods/uniform.py:17-22)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ... import _native as nat
from .apodmethod import ApodizationMethod


@dataclass
class Uniform(ApodizationMethod):
    value: float = 1.0

    def kernel_args(self):
        return nat.APOD_UNIFORM, float(self.value), 0.0

    def calc_apodization(self, arr, target, params=None, transform=None):
        # geometry-independent: a constant vector needs no launch (uniform.py:21-22)
        return np.full(arr.numelements(), self.value)
