"""Simulation grid definition (mirror of openlifu.sim.sim_setup.SimSetup, sim/sim_setup.py:21-230):
extents + spacing -> coordinate vectors; the voxel grid kernel 2 sweeps."""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np

from ..util import dataset as ds
from ..util.dict_conversion import DictMixin
from ..util.units import getunitconversion, getunittype

COORD_DIMS = ("x", "y", "z")
COORD_NAMES = ("Lateral", "Elevation", "Axial")


def _snap(name: str, extent, spacing: float):
    """sim_setup.py:91-105: hi <- lo + round((hi - lo)/spacing) * spacing, warn when far off."""
    n = np.diff(extent) / spacing
    snapped = tuple(np.arange(2) * np.round(n) * spacing + extent[0])
    if ((0.5 - np.abs((n % 1) - 0.5)) / np.round(n)) > 1e-3:
        logging.warning(f"{name} {extent} does not evenly divide by spacing ({spacing}). Rounding to {snapped}.")
    return snapped


def _number(name, v, positive=None):
    if not isinstance(v, (int, float)):
        raise TypeError(f"{name} must be a number.")
    if positive is True and v <= 0:
        raise ValueError(f"{name} must be a positive number.")
    if positive is False and v < 0:
        raise ValueError(f"{name} must be a non-negative number.")


@dataclass
class SimSetup(DictMixin):
    spacing: float = 1.0
    units: str = "mm"
    x_extent: Tuple[float, float] = (-30.0, 30.0)
    y_extent: Tuple[float, float] = (-30.0, 30.0)
    z_extent: Tuple[float, float] = (-4.0, 60.0)
    dt: float = 0.0
    t_end: float = 0.0
    c0: float = 1500.0
    cfl: float = 0.5
    options: dict = field(default_factory=dict)

    def __post_init__(self):
        for name in ("x_extent", "y_extent", "z_extent"):
            e = getattr(self, name)
            if len(e) != 2:
                raise ValueError(f"{name} must have length 2.")
            if e[0] >= e[1]:
                raise ValueError(f"{name} must be in the form (min, max) with min < max.")
        _number("spacing", self.spacing, positive=True)
        if not isinstance(self.units, str):
            raise TypeError("units must be a string.")
        if getunittype(self.units) != "distance":
            raise ValueError(f"units must be a length unit, got {self.units}.")
        _number("c0", self.c0, positive=True)
        _number("cfl", self.cfl, positive=True)
        _number("dt", self.dt, positive=False)
        _number("t_end", self.t_end, positive=False)
        for name in ("x_extent", "y_extent", "z_extent"):
            setattr(self, name, _snap(name, getattr(self, name), self.spacing))

    def _extents(self):
        return [self.x_extent, self.y_extent, self.z_extent]

    def get_size(self, dims=None):
        """n = round(diff/spacing) + 1 per axis (sim_setup.py:152-155)."""
        dims = COORD_DIMS if dims is None else dims
        n = [int(np.round(np.diff(e) / self.spacing).item()) + 1 for e in self._extents()]
        return np.array([n[COORD_DIMS.index(d)] for d in dims]).squeeze()

    def get_extent(self, dims=None, units: str | None = None):
        dims = COORD_DIMS if dims is None else dims
        scl = getunitconversion(self.units, self.units if units is None else units)
        e = self._extents()
        return np.array([e[COORD_DIMS.index(d)] for d in dims]) * scl

    def get_spacing(self, units: str | None = None):
        return getunitconversion(self.units, self.units if units is None else units) * self.spacing

    def get_corners(self, units: str | None = None):
        scl = getunitconversion(self.units, self.units if units is None else units)
        xyz = np.array(np.meshgrid(self.x_extent, self.y_extent, self.z_extent, indexing="ij"))
        return xyz.reshape(3, -1) * scl

    def get_coords(self, dims=None, units: str | None = None):
        """linspace(lo, hi, n) per axis with attrs units / long_name (sim_setup.py:107-116)."""
        dims = COORD_DIMS if dims is None else dims
        units = self.units if units is None else units
        sizes = np.atleast_1d(self.get_size(dims))
        ext = self.get_extent(dims, units)
        return ds.make_coords({d: np.linspace(ext[i][0], ext[i][1], int(sizes[i])) for i, d in enumerate(dims)},
                              {d: {"units": units, "long_name": COORD_NAMES[COORD_DIMS.index(d)]} for d in dims})

    def get_max_cycle_offset(self, arr, frequency: float | None = None, delays=None, zmin: float = 10e-3):
        """Largest spread of element times of flight over the grid points with z >= zmin [m], in cycles
        (sim/sim_setup.py:132-143); the V x N pair loop runs on the device (`tof_spread_k`, fp64)."""
        from .. import get_engine
        frequency = arr.frequency if frequency is None else frequency
        coords = self.get_coords(units="m")
        xs, ys, zs = (np.asarray(c.data, dtype=np.float64) for c in coords.values())
        zs = zs[zs >= zmin]
        if zs.size == 0:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")  # np.max of an empty grid
        eng = get_engine()
        eng.bind(arr)
        return eng.ctx.tof_spread(xs, ys, zs, delays, self.c0) * frequency

    def get_max_distance(self, arr, units: str | None = None):
        """Largest element-to-grid-corner distance (sim_setup.py:145-150), vectorised."""
        units = self.units if units is None else units
        corners = self.get_corners(units=units).T
        pos = arr.get_positions(units=units)
        return float(np.sqrt(((pos[:, None, :] - corners[None, :, :]) ** 2).sum(axis=2)).max())

    def setup_sim_scene(self, seg_method, volume=None):
        """params Dataset for the grid (sim_setup.py:161-188)."""
        if volume is None:
            return seg_method.ref_params(self.get_coords())
        return seg_method.seg_params(volume)

    @staticmethod
    def from_dict(d: dict, on_keyword_mismatch="warn") -> "SimSetup":
        if not isinstance(d, dict):
            raise TypeError("Input must be a dictionary.")
        expected = ["spacing", "units", "x_extent", "y_extent", "z_extent", "dt", "t_end", "c0", "cfl", "options"]
        unexpected = [k for k in d if k not in expected]
        if unexpected:
            if on_keyword_mismatch == "raise":
                raise TypeError(f"Unexpected keyword arguments for SimSetup: {unexpected}")
            if on_keyword_mismatch == "warn":
                logging.warning(f"Ignoring unexpected keyword arguments for SimSetup: {unexpected}")
        return SimSetup(**{k: v for k, v in d.items() if k in expected})
