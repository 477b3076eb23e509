"""Uniform media (mirror of openlifu.seg.seg_methods.uniform, seg/seg_methods/uniform.py:10-65)."""
from __future__ import annotations

from ..material import MATERIALS, Material
from ..seg_method import SegmentationMethod


class UniformSegmentation(SegmentationMethod):
    def _segment(self, volume):
        return self._ref_segment(volume.coords)


class _FixedReference(UniformSegmentation):
    _REF = "water"

    def __init__(self, materials: dict[str, Material] | None = None):
        super().__init__(materials=MATERIALS.copy() if materials is None else materials, ref_material=self._REF)

    def to_dict(self):
        d = super().to_dict()
        d.pop("ref_material")
        return d


class UniformTissue(_FixedReference):
    """Every voxel is tissue."""
    _REF = "tissue"


class UniformWater(_FixedReference):
    """Every voxel is water."""
    _REF = "water"
