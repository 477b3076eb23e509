"""Label volume -> per-voxel medium parameters (mirror of openlifu.seg.seg_method, seg/seg_method.py:19-125).
``ref_params`` builds the ``params`` Dataset that beamforming and ``run_simulation`` read: one volume per
entry of ``PARAM_INFO`` with attrs units / long_name / ref_value (the reference material's value)."""
from __future__ import annotations

import copy
import inspect
import logging
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from ..util import dataset as ds
from ..util.plugin import lookup
from .material import MATERIALS, PARAM_INFO, Material


@dataclass
class SegmentationMethod(ABC):
    materials: dict = field(default_factory=lambda: MATERIALS.copy())
    ref_material: str = "water"

    def __post_init__(self):
        self.materials = MATERIALS.copy() if self.materials is None else self.materials
        if not isinstance(self.materials, dict):
            raise TypeError(f"Materials must be a dictionary, got {type(self.materials).__name__}.")
        if any(not isinstance(m, Material) for m in self.materials.values()):
            raise TypeError("All materials must be instances of Material class.")
        if self.ref_material not in self.materials:
            raise ValueError(f"Reference material {self.ref_material} not found.")

    @abstractmethod
    def _segment(self, volume):
        """integer label volume on the coordinates of ``volume``."""

    # ---- (de)serialisation ---------------------------------------------------------------------
    def to_dict(self) -> dict[str, Any]:
        out = dict(self.__dict__)
        out["materials"] = {name: m.to_dict() for name, m in self.materials.items()}
        out["class"] = type(self).__name__
        return out

    @staticmethod
    def from_dict(d: dict, on_keyword_mismatch="warn") -> "SegmentationMethod":
        if not isinstance(d, dict):
            raise TypeError(f"Expected dict for from_dict, got {type(d).__name__}")
        cls, kwargs = lookup(__package__ + ".seg_methods", copy.deepcopy(d))
        if kwargs.get("materials") is not None:
            kwargs["materials"] = {k: m if isinstance(m, Material) else Material.from_dict(m)
                                   for k, m in kwargs["materials"].items()}
        accepted = {p.name for p in inspect.signature(cls).parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD}
        extra = [k for k in kwargs if k not in accepted]
        if extra and on_keyword_mismatch == "raise":
            raise TypeError(f"Unexpected keyword arguments for {cls.__name__}: {extra}")
        if extra and on_keyword_mismatch == "warn":
            logging.warning(f"Ignoring unexpected keyword arguments for {cls.__name__}: {extra}")
        return cls(**{k: v for k, v in kwargs.items() if k in accepted})

    # ---- labels -> parameter volumes ---------------------------------------------------------------
    def _material_indices(self, materials: dict | None = None):
        return {name: i for i, name in enumerate((self.materials if materials is None else materials))}

    def _map_params(self, seg, materials: dict | None = None):
        """One volume per parameter: vol[label == index(material)] = material.<param> (seg_method.py:84-97)."""
        materials = self.materials if materials is None else materials
        index_of = self._material_indices(materials)
        reference = materials[self.ref_material]
        labels = np.asarray(seg.data)
        volumes = {}
        for pid, info in PARAM_INFO.items():
            lut = np.zeros(max(index_of.values()) + 1)
            for name, mat in materials.items():
                lut[index_of[name]] = getattr(mat, pid)
            known = (labels >= 0) & (labels < lut.size)
            vol = np.where(known, lut[np.clip(labels, 0, lut.size - 1)], 0.0)
            volumes[pid] = ds.make_dataarray(vol, coords=seg.coords, dims=seg.dims, name=pid,
                                             attrs={"units": info["units"], "long_name": info["name"],
                                                    "ref_value": reference.get_param(pid)})
        params = ds.make_dataset(volumes)
        params.attrs["ref_material"] = reference
        return params

    def seg_params(self, volume, materials: dict | None = None):
        return self._map_params(self._segment(volume), materials=self.materials if materials is None else materials)

    def ref_params(self, coords, _internal: bool = False):
        """Uniform reference medium on ``coords`` (seg_method.py:104-107).  Every voxel of every parameter volume holds the
        reference material's value, so the volumes are declared constant and only allocated if somebody reads them
        (same values, shapes, dims and attrs as ``_map_params(_ref_segment(coords))``)."""
        if ds.HAVE_XARRAY and not _internal:  # pragma: no cover - real xarray objects cannot defer
            return self._map_params(self._ref_segment(coords))
        dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
        shape = [len(coords[d]) for d in dims]
        reference = self.materials[self.ref_material]
        cmap = {d: coords[d] for d in dims}
        volumes = {pid: ds.LazyDataArray.uniform(shape, float(getattr(reference, pid)), coords=cmap, dims=dims, name=pid,
                                                 attrs={"units": info["units"], "long_name": info["name"],
                                                        "ref_value": reference.get_param(pid)})
                   for pid, info in PARAM_INFO.items()}
        # (_internal: calc_solution's working copy -- the stand-in Dataset whatever the factories hand out)
        params = ds.Dataset(volumes) if _internal else ds.make_dataset(volumes)
        params.attrs["ref_material"] = reference
        return params

    def _ref_segment(self, coords):
        dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
        shape = [len(coords[d]) for d in dims]
        label = self._material_indices()[self.ref_material]
        return ds.make_dataarray(np.full(shape, label, dtype=int), coords=coords, dims=dims)
