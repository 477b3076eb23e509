"""Single focus (mirror of bf/focal_patterns/single.py:11-33)."""
from __future__ import annotations

from dataclasses import dataclass

from .focal_pattern import FocalPattern


@dataclass
class SinglePoint(FocalPattern):
    def get_targets(self, target):
        return [target.copy()]

    def num_foci(self):
        return 1
