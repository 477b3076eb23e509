"""Medium description: material table and label-volume -> parameter-volume mapping."""
from __future__ import annotations

from . import material as _material
from . import seg_method as _seg_method
from . import seg_methods

Material = _material.Material
MATERIALS = _material.MATERIALS
WATER, TISSUE, SKULL, AIR, STANDOFF = (_material.WATER, _material.TISSUE, _material.SKULL, _material.AIR,
                                       _material.STANDOFF)
SegmentationMethod = _seg_method.SegmentationMethod

__all__ = ("Material", "MATERIALS", "WATER", "TISSUE", "SKULL", "AIR", "STANDOFF", "SegmentationMethod", "seg_methods")
