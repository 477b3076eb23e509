#!/usr/bin/env python3
"""Developer tool (GPU box): kernel ms per launch of BASELINE's 16 x 16 array on an arbitrary grid -- 8-focus shard and single focus, planner's choice and opted out:
  python tools/time_grid.py NXY NZ Z0_MM SPACING_MM [NXY NZ Z0_MM SPACING_MM ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402,F401
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat, dist as od  # noqa: E402

SENS, C0, F0, RHO0 = 1e5, 1500.0, 400e3, 1000.0
eng = ol.get_engine(0); ctx = eng.ctx
arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
eng.bind(arr)
wheel = ol.focal_patterns.Wheel(center=True, num_spokes=63, spoke_radius=5.0)
sweep = np.array([f.get_position(units="m") for f in wheel.get_targets(ol.Point(position=(0, 0, 40), units="mm"))])
shard = sweep[od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]]
a = sys.argv[1:]
for q in range(0, len(a), 4):
    nxy, nz, z0, sp = int(a[q]), int(a[q + 1]), float(a[q + 2]), float(a[q + 3])
    origin = (-(nxy - 1) / 2 * sp * 1e-3, -(nxy - 1) / 2 * sp * 1e-3, z0 * 1e-3)
    for foci, fn in ((shard, "f8"), (sweep[:1] * [0, 0, 1], "f1")):
        ctx.bf_solve(np.asarray(foci), C0)
        for flags, tag in ((nat.OUT_PMAG | nat.OUT_INTENSITY, "auto"), (nat.OUT_PMAG | nat.OUT_INTENSITY | nat.FIELD_FP16_CORRECTION, "fp16")):
            ctx.field_plan(origin, (sp * 1e-3,) * 3, (nxy, nxy, nz), F0, C0, RHO0, SENS, flags=flags)
            for _ in range(5):
                ctx.field_launch()
            ctx.sync()
            ctx.profile_begin(50)
            for _ in range(50):
                ctx.field_launch()
            ctx.sync()
            ms = float(np.mean(ctx.profile_end()))
            vox = nxy * nxy * nz * len(foci)
            print(f"{nxy}x{nxy}x{nz} z0={z0} h={sp} {fn} {tag:5s} {ms:8.4f} ms  {8.0 * vox / ms / 1e6 / 8000 * 100:5.1f} % of 8 TB/s  {ctx.field_variant()[:110]}", flush=True)
