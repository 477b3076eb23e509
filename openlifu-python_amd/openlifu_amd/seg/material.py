"""Acoustic / thermal material table (mirror of openlifu.seg.material, seg/material.py:8-131)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

PARAM_INFO = {
    "sound_speed": {"id": "sound_speed", "name": "Speed of Sound", "units": "m/s"},
    "density": {"id": "density", "name": "Density", "units": "kg/m^3"},
    "attenuation": {"id": "attenuation", "name": "Attenuation", "units": "dB/cm/MHz"},
    "specific_heat": {"id": "specific_heat", "name": "Specific Heat", "units": "J/kg/K"},
    "thermal_conductivity": {"id": "thermal_conductivity", "name": "Thermal Conductivity", "units": "W/m/K"},
}

_CHECKS = (("sound_speed", "Sound speed", True), ("density", "Density", True), ("attenuation", "Attenuation", False),
           ("specific_heat", "Specific heat", True), ("thermal_conductivity", "Thermal conductivity", True))


@dataclass
class Material:
    name: str = "Material"
    sound_speed: float = 1500.0          # m/s
    density: float = 1000.0              # kg/m^3
    attenuation: float = 0.0             # dB/cm/MHz
    specific_heat: float = 4182.0        # J/kg/K
    thermal_conductivity: float = 0.598  # W/m/K

    def __post_init__(self):
        if not isinstance(self.name, str):
            raise TypeError("Material name must be a string.")
        for attr, label, strictly_positive in _CHECKS:
            v = getattr(self, attr)
            if not isinstance(v, (int, float)):
                raise TypeError(f"{label} of {self.name} must be a number.")
            if (v <= 0) if strictly_positive else (v < 0):
                raise ValueError(f"{label} of {self.name} must be {'positive' if strictly_positive else 'non-negative'}.")

    def to_dict(self):
        return {"name": self.name, **{k: getattr(self, k) for k in PARAM_INFO}}

    @classmethod
    def param_info(cls, param_id: str):
        if param_id not in PARAM_INFO:
            raise ValueError(f"Parameter {param_id} not found.")
        return PARAM_INFO[param_id]

    def get_param(self, param_id: str):
        if param_id not in PARAM_INFO:
            raise ValueError(f"Parameter {param_id} not found.")
        return getattr(self, param_id)

    @staticmethod
    def from_dict(d: dict[str, Any]):
        return Material(**d)


WATER = Material("water", 1500.0, 1000.0, 0.0, 4182.0, 0.598)
TISSUE = Material("tissue", 1540.0, 1000.0, 0.0, 3600.0, 0.5)
SKULL = Material("skull", 4080.0, 1900.0, 0.0, 1100.0, 0.3)
AIR = Material("air", 344.0, 1.25, 0.0, 1012.0, 0.025)
STANDOFF = Material("standoff", 1420.0, 1000.0, 1.0, 4182.0, 0.598)
MATERIALS = {"water": WATER, "tissue": TISSUE, "skull": SKULL, "air": AIR, "standoff": STANDOFF}
