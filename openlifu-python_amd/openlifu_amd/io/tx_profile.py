"""Per-focus TX profile numbers of a Solution, quantised on the device (kernel `bf_quantize_k`).

Restates, with citations into /root/reference/src/openlifu/io/LIFUTXDevice.py:
  * delay count      int(delay * getunitconversion(units, 's') * bf_clk)      (:1874, DELAY_WIDTH = 13 bits :78,
                     DEFAULT_CLK_FREQ = 10 MHz :100; a value that does not fit raises ValueError, :1500-1501)
  * apodization bit  1 - apod per channel                                       (:1811)
  * pulse profile    cycles = int(duration * frequency),
                     duty_cycle = 0.66 * max(apod) * amplitude                  (:1358-1364, DEFAULT_PATTERN_DUTY_CYCLE :81)
  * more than one focus is refused by the device driver today (:1355-1356); the numbers are still produced
    per focus so that a caller can program profile after profile.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

DELAY_WIDTH = 13
DEFAULT_CLK_FREQ = 10e6
DEFAULT_PATTERN_DUTY_CYCLE = 0.66


@dataclass
class TxProfile:
    profile: int              # 1-based profile index (delay profile = pulse profile = focus + 1, :1357-1371)
    frequency: float          # Hz
    cycles: int
    duty_cycle: float
    delay_ticks: np.ndarray   # uint16 [N], beamformer-clock counts
    apod_off: np.ndarray      # uint8 [N], 1 = channel disabled


def tx_profiles(solution, bf_clk: float = DEFAULT_CLK_FREQ, engine=None) -> List[TxProfile]:
    """Quantise `solution.delays` / `solution.apodizations` on the device and return one TxProfile per focus.
    Raises ValueError (like set_register_value) when a delay does not fit DELAY_WIDTH bits."""
    from .. import get_engine
    eng = engine or get_engine()
    ctx = eng.ctx
    delays = np.atleast_2d(np.asarray(solution.delays, dtype=np.float64))
    apod = np.atleast_2d(np.asarray(solution.apodizations, dtype=np.float64))
    if delays.shape != apod.shape:
        raise ValueError("Delays and apodizations must have the same number of rows")   # LIFUTXDevice.py:1353-1354
    if ctx.n_el != delays.shape[1]:
        # the steering table needs an element table of matching length; positions are irrelevant for the hand-off
        n = delays.shape[1]
        ctx.set_elements(np.zeros((n, 3)), np.tile([0.0, 0.0, 1.0], (n, 1)), np.ones(n))
        eng._table_key = None   # the bound transducer table was replaced
    ctx.set_steering(delays, apod)
    ticks, aoff, amax, ovf = ctx.bf_quantize(bf_clk, DELAY_WIDTH)
    if ovf.any():
        f = int(np.flatnonzero(ovf)[0])
        bad = int(np.max(np.trunc(delays[f] * bf_clk)))
        raise ValueError(f"Value {bad} does not fit in {DELAY_WIDTH} bits")
    pulse = solution.pulse
    out = []
    for f in range(delays.shape[0]):
        out.append(TxProfile(profile=f + 1, frequency=pulse.frequency, cycles=int(pulse.duration * pulse.frequency),
                             duty_cycle=DEFAULT_PATTERN_DUTY_CYCLE * float(amax[f]) * pulse.amplitude,
                             delay_ticks=ticks[f].copy(), apod_off=aoff[f].copy()))
    return out
