"""Pass / warn / fail limits on analysis outputs (mirror of openlifu.plan.param_constraint.ParameterConstraint,
plan/param_constraint.py:17-100).  A value is fine while ``value <operator> limit`` HOLDS; it is a warning or an
error when the comparison against ``warning_value`` / ``error_value`` fails."""
from __future__ import annotations

import operator as _op
from dataclasses import dataclass

from ..util.dict_conversion import DictMixin

_SCALAR = {"<": _op.lt, "<=": _op.le, ">": _op.gt, ">=": _op.ge}
_RANGE = {
    "within": lambda v, lo, hi: lo < v < hi,
    "inside": lambda v, lo, hi: lo <= v <= hi,
    "outside": lambda v, lo, hi: v < lo or v > hi,
    "outside_inclusive": lambda v, lo, hi: v <= lo or v >= hi,
}


@dataclass
class ParameterConstraint(DictMixin):
    operator: str
    warning_value: float | int | tuple | None = None
    error_value: float | int | tuple | None = None

    def __post_init__(self):
        if self.warning_value is None and self.error_value is None:
            raise ValueError("At least one of warning_value or error_value must be set")
        for label, v in (("Warning", self.warning_value), ("Error", self.error_value)):
            if self.operator in _RANGE:
                if v and (not isinstance(v, tuple) or len(v) != 2 or v[0] >= v[1]):
                    raise ValueError(f"{label} value must be a sorted tuple of two numbers")
            elif self.operator in _SCALAR:
                if v is not None and not isinstance(v, (int, float)):
                    raise ValueError(f"{label} value must be a single value")

    @staticmethod
    def compare(value, operator, threshold) -> bool:
        if operator in _SCALAR:
            return bool(_SCALAR[operator](value, threshold))
        if operator in _RANGE:
            return bool(_RANGE[operator](value, threshold[0], threshold[1]))
        raise ValueError(f"Unsupported operator: {operator}")

    def is_warning(self, value) -> bool:
        return self.warning_value is not None and not self.compare(value, self.operator, self.warning_value)

    def is_error(self, value) -> bool:
        return self.error_value is not None and not self.compare(value, self.operator, self.error_value)

    # Display helpers of the reference (plan/param_constraint.py:11, 81-98: PARAM_STATUS_SYMBOLS, get_status_symbol, to_table -> pandas) are
    # outside this build's scope (no arithmetic on the path).  They exist as explicit refusals so that a caller
    # written against the reference fails with a message instead of an AttributeError; get_status / is_warning / is_error carry the logic.
    def get_status_symbol(self, value) -> str:
        raise NotImplementedError("ParameterConstraint.get_status_symbol (display glyphs) is not part of openlifu_amd; use get_status(value) -> 'ok' | 'warning' | 'error'")

    def to_table(self):
        raise NotImplementedError("ParameterConstraint.to_table (pandas display table) is not part of openlifu_amd; the limits are warning_value / error_value, "
                                  "the checks is_warning / is_error / get_status")

    def get_status(self, value) -> str:
        return "error" if self.is_error(value) else "warning" if self.is_warning(value) else "ok"

    @classmethod
    def from_dict(cls, parameter_dict):
        d = {k: v for k, v in parameter_dict.items() if k != "class"}
        for k in ("warning_value", "error_value"):  # JSON turns the (lo, hi) tuples into lists
            if isinstance(d.get(k), list):
                d[k] = tuple(d[k])
        return cls(**d)
