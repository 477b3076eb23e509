"""Multi-module arrays (mirror of openlifu.xdc.transducerarray, xdc/transducerarray.py:13-137).
``to_transducer`` flattens the modules into one Transducer whose element order,
pins and indices match the reference (pins/indices offset by the running count)."""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

from ..util.units import getunitconversion
from .transducer import Transducer, TransformedTransducer


def get_angle_from_gap(width, gap, roc):
    """Half-angle subtended by one module of a concave cylinder (transducerarray.py:13-21)."""
    mag = np.hypot(roc, width / 2)
    dth = np.arcsin((gap / 2) / mag) + np.arcsin((width / 2) / mag)
    return dth if roc / mag >= 0 else -dth


def get_roc_from_angle(width, gap, dth):
    return (0.5 * gap + 0.5 * width * np.cos(dth)) / np.sin(dth)


@dataclass
class TransducerArray:
    id: str = "transducer_array"
    name: str = "Transducer Array"
    modules: list = field(default_factory=list)
    attrs: dict = field(default_factory=dict)

    def to_transducer(self, offset_pins=True, offset_indices=True) -> Transducer:
        t = Transducer.merge([m.bake() for m in self.modules], offset_pins=offset_pins,
                             offset_indices=offset_indices, merged_attrs=self.attrs)
        t.name = self.name
        t.id = self.id
        return t

    @staticmethod
    def from_dict(data: dict):
        d = {k: v for k, v in data.items() if k != "type"}
        d["modules"] = [TransformedTransducer.from_dict(t) for t in data["modules"]]
        for key in ("standoff_transform", "impulse_response"):
            if d.get("attrs", {}).get(key) is not None:
                d["attrs"][key] = np.array(d["attrs"][key])
        return TransducerArray(**d)

    def to_dict(self):
        d = {"type": "TransducerArray", "id": self.id, "name": self.name,
             "modules": [t.to_dict() for t in self.modules],
             "attrs": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in self.attrs.items()}}
        return d

    def to_json(self, compact: bool = False) -> str:
        return json.dumps(self.to_dict(), separators=(",", ":")) if compact else json.dumps(self.to_dict(), indent=4)

    def to_file(self, file_path: str, compact: bool = False) -> None:
        with open(file_path, "w") as f:
            f.write(self.to_json(compact=compact))

    @staticmethod
    def from_file(filename: str) -> "TransducerArray":
        with open(filename) as f:
            return TransducerArray.from_dict(json.load(f))

    @staticmethod
    def get_concave_cylinder(trans, rows=1, cols=1, width=40, gap=0, dth=None, roc=np.inf, units="mm",
                             id="transducer_array", name="Transducer Array", attrs: dict | None = None):
        """rows x cols modules on a plane (roc = inf) or on a cylinder of radius roc about y
        (transducerarray.py:86-115).  Module placement M maps module -> array coordinates; the
        stored transform is inv(M)."""
        scl = getunitconversion(units, trans.units)
        modules = []
        flat = roc == np.inf
        if not flat and dth is None:
            dth = get_angle_from_gap(width, gap, roc)
        for i in range(rows):
            y = (width + gap) * (i - (rows - 1) / 2) * scl
            for j in range(cols):
                M = np.eye(4)
                M[1, 3] = y
                if flat:
                    M[0, 3] = (width + gap) * (j - (cols - 1) / 2) * scl
                else:
                    th = dth * 2 * (j - (cols - 1) / 2)
                    M[0, 0] = np.cos(th); M[0, 2] = -np.sin(th)
                    M[2, 0] = np.sin(th); M[2, 2] = np.cos(th)
                    M[0, 3] = roc * np.sin(th) * scl
                    M[2, 3] = roc * (1 - np.cos(th)) * scl
                modules.append(TransformedTransducer.from_transducer(trans, transform=np.linalg.inv(M)))
        return TransducerArray(modules=modules, id=id, name=name, attrs=attrs or {})
