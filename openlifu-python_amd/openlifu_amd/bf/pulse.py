"""Sinusoidal pulse (mirror of openlifu.bf.pulse.Pulse, bf/pulse.py:13-63): frequency and amplitude feed
the field kernel (amplitude * voltage * sensitivity = surface pressure), duration the duty cycles."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..util.dict_conversion import DictMixin
from ..util.validate import positive


@dataclass
class Pulse(DictMixin):
    frequency: float = 1.0  # Hz
    amplitude: float = 1.0  # AU in [0, 1]
    duration: float = 1.0   # s

    def __post_init__(self):
        positive("Frequency must be greater than 0", self.frequency)
        if not 0 <= self.amplitude <= 1:
            raise ValueError("Amplitude must be between 0 and 1")
        positive("Duration must be greater than 0", self.duration)

    def calc_pulse(self, t):
        return self.amplitude * np.sin(2 * np.pi * self.frequency * t)

    def calc_time(self, dt: float):
        return np.arange(0, self.duration, dt)
