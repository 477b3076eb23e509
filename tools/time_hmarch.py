#!/usr/bin/env python3
"""Developer tool (GPU box): BASELINE configs[4] (skull slab, marched ray sums, kernel 2m) -- ms per launch sequence for 1 and 8 foci, with the
fused writers (default) and with single-plane writers (OLX_MARCH_FUSE=0 / 2 / 3):  python tools/time_hmarch.py [fuse settings...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402,F401
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat, dist as od  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402
from openlifu_amd.seg.seg_methods import skull_slab_volumes  # noqa: E402

SENS, C0, F0, RHO0 = 1e5, 1500.0, 400e3, 1000.0
eng = ol.get_engine(0); ctx = eng.ctx
sf = od.ShardedField(eng, 1, 0)
half = (256 - 1) / 2 * 0.25
setup = ol.SimSetup(spacing=0.25, x_extent=(-half, half), y_extent=(-half, half), z_extent=(5.0, 5.0 + 255 * 0.25))
origin, spacing, n = grid_from_coords(setup.get_coords())
coords = [np.asarray(c.data) * 1e-3 for c in setup.get_coords().values()]
arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
skull = skull_slab_volumes(*coords); skull["model"] = "marched"
focus = np.array([[0.0, 0.0, 40e-3]])
wheel = ol.focal_patterns.Wheel(center=True, num_spokes=63, spoke_radius=5.0)
sweep = np.array([f.get_position(units="m") for f in wheel.get_targets(ol.Point(position=(0, 0, 40), units="mm"))])
shard = sweep[od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]]
for fuse in (sys.argv[1:] or ["default", "0"]):
    if fuse == "default":
        os.environ.pop("OLX_MARCH_FUSE", None)
    else:
        os.environ["OLX_MARCH_FUSE"] = fuse
    for key, foci, steps in (("c5_skull_f1", focus, 30), ("c5_skull_f8", shard, 10)):
        dl, ap = eng.beamform(arr, foci, C0)
        sf.plan_slab_sweep(arr, dl, ap, origin, spacing, n, F0, C0, RHO0, SENS, flags=nat.OUT_PMAG | nat.OUT_INTENSITY, medium=skull)
        for _ in range(3):
            sf.step("none")
        ctx.sync()
        ctx.profile_begin(steps)
        for _ in range(steps):
            sf.step("none")
        ctx.sync()
        ms = ctx.profile_end()
        if key == "c5_skull_f8" and fuse != "default":
            continue
        print(f"OLX_MARCH_FUSE={fuse:8s} TI={os.environ.get('OLX_MARCH_FUSE_TI', '16'):3s} {key:12s} {float(np.mean(ms)):8.4f} ms  {ctx.field_variant()[-90:]}", flush=True)
