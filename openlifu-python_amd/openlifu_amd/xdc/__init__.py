from __future__ import annotations

from .element import Element
from .transducer import Transducer, TransformedTransducer
from .transducerarray import TransducerArray, get_angle_from_gap, get_roc_from_angle

__all__ = ["Element", "Transducer", "TransformedTransducer", "TransducerArray",
           "get_angle_from_gap", "get_roc_from_angle"]
