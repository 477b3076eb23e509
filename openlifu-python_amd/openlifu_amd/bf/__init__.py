"""Beamforming layer of the hot path: pulse / sequence dataclasses plus the three plug-in families
(delay methods, apodization methods, focal patterns)."""
from __future__ import annotations

from . import apod_methods, delay_methods, focal_patterns
from . import pulse as _pulse
from . import sequence as _sequence

Pulse = _pulse.Pulse
Sequence = _sequence.Sequence
DelayMethod = delay_methods.DelayMethod
ApodizationMethod = apod_methods.ApodizationMethod
FocalPattern = focal_patterns.FocalPattern
SinglePoint = focal_patterns.SinglePoint
Wheel = focal_patterns.Wheel

__all__ = ("Pulse", "Sequence", "DelayMethod", "ApodizationMethod", "FocalPattern", "SinglePoint", "Wheel",
           "delay_methods", "apod_methods", "focal_patterns")
