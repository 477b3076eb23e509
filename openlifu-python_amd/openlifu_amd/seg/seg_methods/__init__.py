"""Segmentation plug-ins (class-name lookup namespace for ``SegmentationMethod.from_dict``).
Upstream ships only uniform media; ``ThresholdSegmentation`` / ``SkullThreshold`` produce the label volume the
heterogeneous field kernel consumes."""
from __future__ import annotations

from . import threshold as _threshold
from . import uniform as _uniform

UniformSegmentation = _uniform.UniformSegmentation
UniformWater = _uniform.UniformWater
UniformTissue = _uniform.UniformTissue
ThresholdSegmentation = _threshold.ThresholdSegmentation
SkullThreshold = _threshold.SkullThreshold
skull_slab_image = _threshold.skull_slab_image
skull_slab_volumes = _threshold.skull_slab_volumes

__all__ = ("UniformSegmentation", "UniformWater", "UniformTissue", "ThresholdSegmentation", "SkullThreshold",
           "skull_slab_image", "skull_slab_volumes")
