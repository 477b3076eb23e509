from __future__ import annotations

from . import apod_methods, delay_methods, focal_patterns
from .apod_methods import ApodizationMethod
from .delay_methods import DelayMethod
from .focal_patterns import FocalPattern, SinglePoint, Wheel
from .pulse import Pulse
from .sequence import Sequence

__all__ = ["DelayMethod", "ApodizationMethod", "Wheel", "FocalPattern", "SinglePoint", "Pulse", "Sequence",
           "delay_methods", "apod_methods", "focal_patterns"]
