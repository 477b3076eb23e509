"""dataclass <-> dict helper (mirror of openlifu.util.dict_conversion.DictMixin,
util/dict_conversion.py:11-40: ``to_dict`` = asdict, ``from_dict`` drops a "class"
key and converts ndarray-annotated fields)."""
from __future__ import annotations

from dataclasses import asdict, fields
from typing import Any, Dict

import numpy as np


class DictMixin:
    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @classmethod
    def from_dict(cls, parameter_dict: Dict[str, Any]):
        parameter_dict = {k: v for k, v in parameter_dict.items() if k != "class"}
        obj = cls(**parameter_dict)
        for f in fields(cls):
            if f.type is np.ndarray or (isinstance(f.type, str) and "np.ndarray" in f.type):
                setattr(obj, f.name, np.array(getattr(obj, f.name)))
        return obj
