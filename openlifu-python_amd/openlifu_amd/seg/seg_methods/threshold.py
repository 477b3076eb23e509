"""Threshold segmentation of an image volume (CT in Hounsfield units, or any scalar image already resampled onto the
simulation grid -- SimSetup.setup_sim_scene, sim/sim_setup.py:161-188) into the label volume the medium plumbing of
the reference consumes (SegmentationMethod._map_params, seg/seg_method.py:84-97).

The reference ships only uniform media (seg/seg_methods/uniform.py:10-65); SURVEY 8(f)3 asks for "a threshold-based
CT/MR segmenter producing the label volume" so that the heterogeneous kernel (kernel 2h) has a
producer behind the reference's own class-name lookup (``SegmentationMethod.from_dict``, seg_method.py:46-78):

    {"class": "ThresholdSegmentation", "bounds": [-200.0, 300.0], "labels": ["air", "tissue", "skull"], "ref_material": "water"}
    {"class": "SkullThreshold", "skull_threshold": 300.0}

``bounds`` are ascending image values; a voxel with ``bounds[k-1] <= value < bounds[k]`` gets ``labels[k]``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ...util import dataset as ds
from ..material import MATERIALS, Material
from ..seg_method import SegmentationMethod


@dataclass
class ThresholdSegmentation(SegmentationMethod):
    bounds: list = field(default_factory=lambda: [300.0])
    labels: list = field(default_factory=lambda: ["water", "skull"])

    def __post_init__(self):
        super().__post_init__()
        self.bounds = [float(b) for b in self.bounds]
        self.labels = list(self.labels)
        if len(self.labels) != len(self.bounds) + 1:
            raise ValueError(f"Need len(bounds) + 1 labels, got {len(self.labels)} labels for {len(self.bounds)} bounds.")
        if any(b1 <= b0 for b0, b1 in zip(self.bounds, self.bounds[1:])):
            raise ValueError("Threshold bounds must be strictly ascending.")
        missing = [m for m in self.labels if m not in self.materials]
        if missing:
            raise ValueError(f"Label materials {missing} not found in the material table.")

    def _segment(self, volume):
        index_of = self._material_indices()
        lut = np.array([index_of[m] for m in self.labels], dtype=int)
        seg = lut[np.digitize(np.asarray(volume.data), self.bounds)]
        return ds.make_dataarray(seg, coords=volume.coords, dims=volume.dims)


class SkullThreshold(ThresholdSegmentation):
    """Two-class head model: image value >= ``skull_threshold`` (default 300 HU) is skull, everything else the
    reference material (water)."""

    def __init__(self, skull_threshold: float = 300.0, materials: dict[str, Material] | None = None, ref_material: str = "water"):
        self.skull_threshold = float(skull_threshold)
        super().__init__(materials=MATERIALS.copy() if materials is None else materials, ref_material=ref_material,
                         bounds=[self.skull_threshold], labels=[ref_material, "skull"])

    def to_dict(self):
        d = super().to_dict()
        d.pop("bounds"); d.pop("labels")
        return d


def skull_slab_image(xs_m, ys_m, zs_m, inside: float = 1000.0, outside: float = 0.0) -> np.ndarray:
    """Synthetic "CT" of SURVEY 8(d)'s skull-slab phantom on a grid (axis vectors in metres): voxels with
    8 mm <= z < 14 mm + 2 mm sin(2 pi x / 40 mm) cos(2 pi y / 40 mm) hold ``inside`` (bone, HU), the rest ``outside``."""
    xs, ys, zs = (np.asarray(v, dtype=np.float64) for v in (xs_m, ys_m, zs_m))
    zsurf = 14e-3 + 2e-3 * np.sin(2 * np.pi * xs / 40e-3)[:, None] * np.cos(2 * np.pi * ys / 40e-3)[None, :]
    mask = (zs[None, None, :] >= 8e-3) & (zs[None, None, :] < zsurf[:, :, None])
    return np.where(mask, np.float32(inside), np.float32(outside)).astype(np.float32)


def skull_slab_volumes(xs_m, ys_m, zs_m, c_skull=2800.0, rho_skull=1900.0, alpha_skull=6.0):
    """The medium volumes of BASELINE configs[4] / SURVEY 8(d) (skull c 2800 m/s, rho 1900 kg/m^3, alpha 6 dB/cm/MHz per
    tests/resources/example_db/protocols/example_protocol/example_protocol.json:49-56; water elsewhere), made the way the
    product makes them: threshold segmentation of the synthetic image, then the reference's label -> parameter map."""
    materials = MATERIALS.copy()
    materials["skull"] = Material("skull", float(c_skull), float(rho_skull), float(alpha_skull), 1100.0, 0.3)
    seg = SkullThreshold(300.0, materials=materials)
    coords = {"x": np.asarray(xs_m), "y": np.asarray(ys_m), "z": np.asarray(zs_m)}
    img = ds.make_dataarray(skull_slab_image(xs_m, ys_m, zs_m), coords=coords, dims=("x", "y", "z"))
    params = seg.seg_params(img)
    return {k: np.asarray(params[k].data, dtype=np.float32) for k in ("sound_speed", "attenuation", "density")}
