from __future__ import annotations

from . import dataset, units
from .dict_conversion import DictMixin

__all__ = ["units", "dataset", "DictMixin"]
