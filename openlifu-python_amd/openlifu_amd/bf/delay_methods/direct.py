"""Geometric time-of-flight delays (mirror of bf/delay_methods/direct.py:16-38), computed by
HIP kernel 1 (``bf_solve_k``) instead of a Python loop over elements."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ...engine import get_engine
from .delaymethod import DelayMethod


@dataclass
class Direct(DelayMethod):
    c0: float = 1480.0  # m/s, used only when no params are given (direct.py:29-32)

    def __post_init__(self):
        if not isinstance(self.c0, (int, float)):
            raise TypeError("Speed of sound must be a number")
        if self.c0 <= 0:
            raise ValueError("Speed of sound must be greater than 0")
        self.c0 = float(self.c0)

    def speed(self, params) -> float:
        return self.c0 if params is None else float(params["sound_speed"].attrs["ref_value"])

    def calc_delays(self, arr, target, params=None, transform: np.ndarray | None = None):
        """delays[N] [s] = max(tof) - tof for one focus.  A list of Points returns [F,N]."""
        delays, _ = get_engine().beamform(arr, target, self.speed(params), transform=transform)
        return delays if isinstance(target, (list, tuple)) else delays[0]
