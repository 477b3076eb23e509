"""``Point`` -- the focus / target container (mirror of openlifu.geo.Point,
geo.py:17-154).  Rendering helpers (vtk) are out of scope."""
from __future__ import annotations

import copy
import json
from dataclasses import dataclass, field
from typing import Any, Dict, Tuple

import numpy as np

from .util.units import getunitconversion


@dataclass
class Point:
    position: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0]))
    id: str = "point"
    name: str = "Point"
    color: Any = (1.0, 0.0, 0.0)
    radius: float = 1.0
    dims: Tuple[str, str, str] = ("x", "y", "z")
    units: str = "mm"

    def __post_init__(self):
        if len(self.position) != len(self.dims):
            raise ValueError("Position and dims must have same length.")
        self.position = np.array(self.position).reshape(3)

    def copy(self) -> "Point":
        """An independent Point (geo.py: ``copy.deepcopy(self)``).  Field by field: the position array and a mutable colour are copied, the rest
        is immutable -- a focal pattern copies every focus of every call, and deepcopy's generic walk was 25 calls per Point."""
        new = object.__new__(type(self))
        d = dict(self.__dict__)
        d["position"] = np.array(self.position)
        for k, v in d.items():
            if k != "position" and not isinstance(v, (str, int, float, bool, tuple, type(None))):
                d[k] = copy.deepcopy(v)
        new.__dict__ = d
        return new

    def get_position(self, dim=None, units: str | None = None):
        """geo.py:48-54."""
        scl = getunitconversion(self.units, self.units if units is None else units)
        if dim is None:
            return self.position * scl
        return self.position[self.dims.index(dim)] * scl

    def get_matrix(self, origin: np.ndarray | None = None, center_on_point: bool = True, local: bool = False):
        """Focal frame of the point (geo.py:56-74): z axis along the ray origin->point,
        x axis in the x-z plane, y = z cross x; translation = the point (or zero)."""
        origin = np.eye(4) if origin is None else np.asarray(origin, dtype=float)
        pos = (np.linalg.inv(origin) @ np.append(self.position, 1.0))[:3]
        r = np.linalg.norm(pos)
        zvec = pos / r if r != 0 else np.array([0.0, 0.0, 1.0])
        az = -np.arctan2(zvec[0], zvec[2])
        xvec = np.array([np.cos(az), 0.0, np.sin(az)])
        m = np.eye(4)
        m[:3, 0] = xvec
        m[:3, 1] = np.cross(zvec, xvec)
        m[:3, 2] = zvec
        m[:3, 3] = pos if center_on_point else 0.0
        return m if local else origin @ m

    def rescale(self, units: str):
        scl = getunitconversion(self.units, units)
        self.position = self.position * scl
        self.radius = self.radius * scl
        self.units = units

    def transform(self, matrix: np.ndarray, units: str | None = None, new_dims=None):
        if units is not None:
            self.rescale(units)
        self.position = (np.asarray(matrix) @ np.append(self.position, 1.0))[:3]
        if new_dims is not None:
            self.dims = new_dims

    def to_dict(self):
        return {"id": self.id, "name": self.name, "color": self.color, "radius": self.radius,
                "position": self.position.tolist(), "dims": self.dims, "units": self.units}

    @staticmethod
    def from_dict(point_data: Dict):
        d = dict(point_data)
        if "color" in d:
            if len(d["color"]) != 3:
                raise ValueError(f"Color should have three components; got {d['color']}.")
            d["color"] = tuple(float(c) for c in d["color"])
        if "radius" in d:
            d["radius"] = float(d["radius"])
        if "position" in d:
            d["position"] = np.array(d["position"])
        if "dims" in d:
            d["dims"] = tuple(d["dims"])
        return Point(**d)

    @staticmethod
    def from_json(json_string: str) -> "Point":
        return Point.from_dict(json.loads(json_string))

    def to_json(self, compact: bool) -> str:
        return json.dumps(self.to_dict(), separators=(",", ":")) if compact else json.dumps(self.to_dict(), indent=4)
