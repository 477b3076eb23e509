"""openlifu_amd -- MI355X-native beamforming + acoustic-field hot path behind openlifu's API.

Drop-in usage:  ``import openlifu_amd as openlifu`` then ``openlifu.Protocol(...).calc_solution(...)``,
``protocol.beamform(...)`` or ``openlifu.sim.run_simulation(...)`` exactly as with the reference
(public names follow src/openlifu/__init__.py:9-72 for the subset on the hot path).
Host code is plain Python + NumPy; all arithmetic on the path runs in hand-written HIP kernels
reached through the C-ABI in include/olx.h (ctypes, no PyTorch).
"""
from __future__ import annotations

from . import bf, geo, plan, seg, sim, util, xdc
from .bf import ApodizationMethod, DelayMethod, FocalPattern, Pulse, Sequence, apod_methods, delay_methods, focal_patterns
from .engine import get_engine, gpu_available
from .geo import Point
from .plan import Protocol, Solution
from .seg import AIR, MATERIALS, SKULL, STANDOFF, TISSUE, WATER, Material, SegmentationMethod, seg_methods
from .sim import SimSetup
from .xdc import Element, Transducer, TransducerArray

__version__ = "0.1.0"

__all__ = ["Point", "Transducer", "TransducerArray", "Element", "Protocol", "Solution", "Material", "SegmentationMethod",
           "seg_methods", "MATERIALS", "WATER", "TISSUE", "SKULL", "AIR", "STANDOFF", "DelayMethod", "ApodizationMethod",
           "Pulse", "Sequence", "FocalPattern", "focal_patterns", "delay_methods", "apod_methods", "SimSetup",
           "bf", "geo", "plan", "seg", "sim", "util", "xdc", "get_engine", "gpu_available"]
