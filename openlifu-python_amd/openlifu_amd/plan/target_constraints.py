"""Target bound checks (mirror of plan/target_constraints.py:14-71); only ``check_bounds`` is on
the calc_solution path (plan/protocol.py:207-224, :302)."""
from __future__ import annotations

import logging
from dataclasses import dataclass

from ..util.dict_conversion import DictMixin
from ..util.units import getunittype


@dataclass
class TargetConstraints(DictMixin):
    dim: str = "x"
    name: str = "dim"
    units: str = "m"
    min: float = float("-inf")
    max: float = float("inf")

    def __post_init__(self):
        for label, v in (("Dimension ID", self.dim), ("Dimension name", self.name), ("Dimension units", self.units)):
            if not isinstance(v, str):
                raise TypeError(f"{label} must be a string")
        if getunittype(self.units) != "distance":
            raise ValueError(f"Units must be a length unit, got {self.units}")
        if not isinstance(self.min, (int, float)):
            raise TypeError("Minimum value must be a number")
        if not isinstance(self.max, (int, float)):
            raise TypeError("Maximum value must be a number")
        if self.min > self.max:
            raise ValueError("Minimum value cannot be greater than maximum value")

    def check_bounds(self, pos: float):
        if pos < self.min or pos > self.max:
            msg = f"The position {pos} at dimension {self.name} is not within bounds [{self.min}, {self.max}]!"
            logging.error(msg=msg)
            raise ValueError(msg)
