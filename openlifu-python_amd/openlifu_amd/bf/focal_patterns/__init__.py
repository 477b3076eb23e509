from __future__ import annotations

from .focal_pattern import FocalPattern
from .single import SinglePoint
from .wheel import Wheel

__all__ = ["FocalPattern", "SinglePoint", "Wheel"]
