"""Shared field validators.  Messages and exception types follow the reference's ``__post_init__``
checks (golden G9 pins the exception types) without repeating each check inline."""
from __future__ import annotations

from .units import getunittype


def number(label: str, value, article_type: bool = True):
    """TypeError('<label> must be a number, got <type>.') unless int/float."""
    if not isinstance(value, (int, float)):
        suffix = f", got {type(value).__name__}." if article_type else ""
        raise TypeError(f"{label} must be a number{suffix}")


def non_negative(label: str, value):
    if value < 0:
        raise ValueError(f"{label} must be non-negative, got {value}.")


def unit_kind(units: str, kind: str, message: str):
    if getunittype(units) != kind:
        raise ValueError(message)


def positive(message: str, value, strict: bool = True):
    if (value <= 0) if strict else (value < 0):
        raise ValueError(message)
