from __future__ import annotations

from .delaymethod import DelayMethod
from .direct import Direct

__all__ = ["DelayMethod", "Direct"]
