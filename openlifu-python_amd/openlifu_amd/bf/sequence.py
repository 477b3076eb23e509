"""Pulse-train timing (mirror of openlifu.bf.sequence.Sequence, bf/sequence.py:12-74)."""
from __future__ import annotations

from dataclasses import dataclass

from ..util.dict_conversion import DictMixin


@dataclass
class Sequence(DictMixin):
    pulse_interval: float = 1.0
    pulse_count: int = 1
    pulse_train_interval: float = 1.0
    pulse_train_count: int = 1

    def __post_init__(self):
        if self.pulse_interval <= 0:
            raise ValueError("Pulse interval must be positive")
        if self.pulse_count <= 0:
            raise ValueError("Pulse count must be positive")
        if self.pulse_train_interval < 0:
            raise ValueError("Pulse train interval must be non-negative")
        if 0 < self.pulse_train_interval < self.pulse_interval * self.pulse_count:
            raise ValueError("Pulse train interval must be greater than or equal to the total pulse interval")
        if self.pulse_train_count <= 0:
            raise ValueError("Pulse train count must be positive")

    def get_pulse_train_duration(self) -> float:
        return self.pulse_interval * self.pulse_count

    def get_sequence_duration(self) -> float:
        interval = self.get_pulse_train_duration() if self.pulse_train_interval == 0 else self.pulse_train_interval
        return interval * self.pulse_train_count
