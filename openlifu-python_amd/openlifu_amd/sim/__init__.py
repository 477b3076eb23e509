"""Simulation grid + the ``run_simulation`` seam (HIP kernel 2 instead of k-Wave)."""
from __future__ import annotations

from . import field
from . import sim_setup as _sim_setup

run_simulation = field.run_simulation
SimSetup = _sim_setup.SimSetup

__all__ = ("SimSetup", "run_simulation", "field")
