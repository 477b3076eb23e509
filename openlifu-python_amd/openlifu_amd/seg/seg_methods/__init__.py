from __future__ import annotations

from .uniform import UniformSegmentation, UniformTissue, UniformWater

__all__ = ["UniformSegmentation", "UniformWater", "UniformTissue"]
