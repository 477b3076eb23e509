"""Delay-method plug-ins.

Namespace used for class-name lookup: ``DelayMethod.from_dict({"class": "Direct", ...})`` resolves
``"Direct"`` here, exactly like the reference's package of the same name.
"""
from __future__ import annotations

from . import delaymethod as _base
from . import direct as _direct

DelayMethod = _base.DelayMethod
Direct = _direct.Direct

__all__ = ("DelayMethod", "Direct")
