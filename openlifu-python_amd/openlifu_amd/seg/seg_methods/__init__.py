"""Segmentation plug-ins (class-name lookup namespace for ``SegmentationMethod.from_dict``).
Only uniform media exist upstream; the field kernels use the reference material of whichever is chosen."""
from __future__ import annotations

from . import uniform as _uniform

UniformSegmentation = _uniform.UniformSegmentation
UniformWater = _uniform.UniformWater
UniformTissue = _uniform.UniformTissue

__all__ = ("UniformSegmentation", "UniformWater", "UniformTissue")
