"""Pulse-train timing (mirror of openlifu.bf.sequence.Sequence, bf/sequence.py:12-74)."""
from __future__ import annotations

from dataclasses import dataclass

from ..util.dict_conversion import DictMixin
from ..util.validate import positive


@dataclass
class Sequence(DictMixin):
    pulse_interval: float = 1.0        # s between pulses
    pulse_count: int = 1
    pulse_train_interval: float = 1.0  # s between trains (0 = back to back)
    pulse_train_count: int = 1

    def __post_init__(self):
        positive("Pulse interval must be positive", self.pulse_interval)
        positive("Pulse count must be positive", self.pulse_count)
        positive("Pulse train interval must be non-negative", self.pulse_train_interval, strict=False)
        train = self.get_pulse_train_duration()
        if 0 < self.pulse_train_interval < train:
            raise ValueError("Pulse train interval must be greater than or equal to the total pulse interval")
        positive("Pulse train count must be positive", self.pulse_train_count)

    def get_pulse_train_duration(self) -> float:
        return self.pulse_interval * self.pulse_count

    def get_sequence_duration(self) -> float:
        per_train = self.pulse_train_interval or self.get_pulse_train_duration()
        return per_train * self.pulse_train_count
