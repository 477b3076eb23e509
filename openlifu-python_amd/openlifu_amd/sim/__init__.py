from __future__ import annotations

from . import field
from .field import run_simulation
from .sim_setup import SimSetup

__all__ = ["SimSetup", "run_simulation", "field"]
