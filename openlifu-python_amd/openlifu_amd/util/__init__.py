"""Unit parsing, the labelled-array stand-in for xarray, dataclass <-> dict helper, validators."""
from __future__ import annotations

from . import dataset, units, validate
from . import dict_conversion as _dc

DictMixin = _dc.DictMixin

__all__ = ("units", "dataset", "validate", "DictMixin")
