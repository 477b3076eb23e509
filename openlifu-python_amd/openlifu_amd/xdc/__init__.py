"""Device model: elements, transducers, multi-module arrays.  ``Transducer.element_table`` is the bridge
to the GPU (SoA arrays uploaded by ``olx_set_elements``)."""
from __future__ import annotations

from . import element as _element
from . import transducer as _transducer
from . import transducerarray as _array
from .util import load_transducer_from_file

Element = _element.Element
Transducer = _transducer.Transducer
TransformedTransducer = _transducer.TransformedTransducer
TransducerArray = _array.TransducerArray
get_angle_from_gap = _array.get_angle_from_gap
get_roc_from_angle = _array.get_roc_from_angle

__all__ = ("Element", "Transducer", "TransformedTransducer", "TransducerArray", "get_angle_from_gap", "get_roc_from_angle",
           "load_transducer_from_file")
