"""Label volume -> per-voxel medium parameters (mirror of openlifu.seg.seg_method,
seg/seg_method.py:19-125).  ``ref_params`` builds the ``params`` Dataset that beamforming and
``run_simulation`` read: five parameter volumes with attrs units / long_name / ref_value."""
from __future__ import annotations

import copy
import inspect
import logging
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from ..util import dataset as ds
from .material import MATERIALS, PARAM_INFO, Material


@dataclass
class SegmentationMethod(ABC):
    materials: dict = field(default_factory=lambda: MATERIALS.copy())
    ref_material: str = "water"

    def __post_init__(self):
        if self.materials is None:
            self.materials = MATERIALS.copy()
        if not isinstance(self.materials, dict):
            raise TypeError(f"Materials must be a dictionary, got {type(self.materials).__name__}.")
        if not all(isinstance(m, Material) for m in self.materials.values()):
            raise TypeError("All materials must be instances of Material class.")
        if self.ref_material not in self.materials:
            raise ValueError(f"Reference material {self.ref_material} not found.")

    @abstractmethod
    def _segment(self, volume):
        ...

    def to_dict(self) -> dict[str, Any]:
        d = self.__dict__.copy()
        d["materials"] = {k: v.to_dict() for k, v in self.materials.items()}
        d["class"] = self.__class__.__name__
        return d

    @staticmethod
    def from_dict(d: dict, on_keyword_mismatch="warn") -> "SegmentationMethod":
        from . import seg_methods
        if not isinstance(d, dict):
            raise TypeError(f"Expected dict for from_dict, got {type(d).__name__}")
        d = copy.deepcopy(d)
        cls = getattr(seg_methods, d.pop("class"))
        if d.get("materials") is not None:
            d["materials"] = {k: v if isinstance(v, Material) else Material.from_dict(v)
                              for k, v in d["materials"].items()}
        expected = [p.name for p in inspect.signature(cls).parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
        unexpected = [k for k in d if k not in expected]
        if unexpected:
            if on_keyword_mismatch == "raise":
                raise TypeError(f"Unexpected keyword arguments for {cls.__name__}: {unexpected}")
            if on_keyword_mismatch == "warn":
                logging.warning(f"Ignoring unexpected keyword arguments for {cls.__name__}: {unexpected}")
            for k in unexpected:
                d.pop(k)
        return cls(**d)

    def _material_indices(self, materials: dict | None = None):
        materials = self.materials if materials is None else materials
        return {mid: i for i, mid in enumerate(materials.keys())}

    def _map_params(self, seg, materials: dict | None = None):
        """seg_method.py:84-97: one volume per parameter, filled per material label."""
        materials = self.materials if materials is None else materials
        idx = self._material_indices(materials)
        ref = materials[self.ref_material]
        labels = np.asarray(seg.data)
        out = {}
        for pid, info in PARAM_INFO.items():
            vol = np.zeros(labels.shape)
            for mid, mat in materials.items():
                vol[labels == idx[mid]] = getattr(mat, pid)
            out[pid] = ds.make_dataarray(vol, coords=seg.coords, dims=seg.dims, name=pid,
                                         attrs={"units": info["units"], "long_name": info["name"],
                                                "ref_value": ref.get_param(pid)})
        params = ds.make_dataset(out)
        params.attrs["ref_material"] = ref
        return params

    def seg_params(self, volume, materials: dict | None = None):
        materials = self.materials if materials is None else materials
        return self._map_params(self._segment(volume), materials=materials)

    def ref_params(self, coords):
        return self._map_params(self._ref_segment(coords))

    def _ref_segment(self, coords):
        dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
        sz = [len(coords[d]) for d in dims]
        label = self._material_indices()[self.ref_material]
        return ds.make_dataarray(np.full(sz, label, dtype=int), coords=coords, dims=dims)
