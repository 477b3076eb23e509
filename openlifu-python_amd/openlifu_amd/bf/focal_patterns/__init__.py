"""Focal patterns: one target -> the list of foci the kernels solve in a single launch
(class-name lookup namespace for ``FocalPattern.from_dict``)."""
from __future__ import annotations

from . import focal_pattern as _base
from . import single as _single
from . import wheel as _wheel

FocalPattern = _base.FocalPattern
SinglePoint = _single.SinglePoint
Wheel = _wheel.Wheel

__all__ = ("FocalPattern", "SinglePoint", "Wheel")
