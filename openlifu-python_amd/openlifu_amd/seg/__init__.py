from __future__ import annotations

from . import seg_methods
from .material import AIR, MATERIALS, SKULL, STANDOFF, TISSUE, WATER, Material
from .seg_method import SegmentationMethod

__all__ = ["Material", "MATERIALS", "WATER", "TISSUE", "SKULL", "AIR", "STANDOFF", "SegmentationMethod", "seg_methods"]
