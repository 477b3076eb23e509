"""Wheel of foci around a target (mirror of bf/focal_patterns/wheel.py:14-73): optional centre
plus ``num_spokes`` points on a circle of ``spoke_radius`` in the target's focal frame.

Reference quirk reproduced, not fixed: spoke positions mix the target's RAW coordinates with
``spoke_radius`` and are labelled ``distance_units`` (wheel.py:54-63), so a target given in metres
yields spokes like (5, 0, 0.05) "mm".  Use same-unit inputs."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ...geo import Point
from .focal_pattern import FocalPattern


@dataclass
class Wheel(FocalPattern):
    center: bool = True
    num_spokes: int = 4
    spoke_radius: float = 1.0
    distance_units: str = "mm"

    def __post_init__(self):
        if not isinstance(self.center, bool):
            raise TypeError(f"Center must be a boolean, got {type(self.center).__name__}.")
        if not isinstance(self.num_spokes, int) or self.num_spokes < 1:
            raise ValueError(f"Number of spokes must be a positive integer, got {self.num_spokes}.")
        if not isinstance(self.spoke_radius, (int, float)) or self.spoke_radius <= 0:
            raise ValueError(f"Spoke radius must be a positive number, got {self.spoke_radius}.")
        super().__post_init__()

    def get_targets(self, target: Point):
        targets = []
        if self.center:
            c = target.copy()
            c.id = f"{target.id} (Center)"
            targets.append(c)
        frame = target.get_matrix(center_on_point=True)
        theta = 2 * np.pi * np.arange(self.num_spokes) / self.num_spokes
        local = np.stack([self.spoke_radius * np.cos(theta), self.spoke_radius * np.sin(theta),
                          np.zeros_like(theta), np.ones_like(theta)], axis=0)
        world = (frame @ local)[:3].T
        for th, pos in zip(theta, world):
            deg = np.rad2deg(th)
            targets.append(Point(id=f"{target.id}_{deg:.0f}deg", name=f"{target.name} ({deg:.0f}°)",
                                 position=pos, units=self.distance_units, radius=target.radius))
        return targets

    def num_foci(self) -> int:
        return int(self.center) + self.num_spokes
