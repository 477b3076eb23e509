#!/usr/bin/env python3
"""Developer tool (GPU box): kernel time per launch of the legs the lattice kernels do NOT serve (kernels 2a / 2b / 2c) -- the tilted two-module
array and the jittered 16 x 16 array of bench.py's config_legs, 1 / 8 / 64 foci at 256^3 -- for same-box A/B runs of library builds:
  OLX_LIB_PATH=openlifu-python_amd/lib/libolx_X.so python tools/time_general.py [legs...]      (default: all)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402,F401
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat, dist as od  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402

SENS, C0, F0, RHO0 = 1e5, 1500.0, 400e3, 1000.0
want = set(sys.argv[1:])
eng = ol.get_engine(0); ctx = eng.ctx
sf = od.ShardedField(eng, 1, 0)
half = (256 - 1) / 2 * 0.25
setup = ol.SimSetup(spacing=0.25, x_extent=(-half, half), y_extent=(-half, half), z_extent=(5.0, 5.0 + 255 * 0.25))
origin, spacing, n = grid_from_coords(setup.get_coords())
wheel = ol.focal_patterns.Wheel(center=True, num_spokes=63, spoke_radius=5.0)
sweep = np.array([f.get_position(units="m") for f in wheel.get_targets(ol.Point(position=(0, 0, 40), units="mm"))])
shard = sweep[od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]]
focus = np.array([[0.0, 0.0, 40e-3]])
halfarr = ol.Transducer.gen_matrix_array(nx=8, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
tilted = ol.TransducerArray.get_concave_cylinder(halfarr, rows=1, cols=2, width=24.0, gap=0.6, roc=80.0, units="mm").to_transducer()
rng = np.random.default_rng(147)
jit = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
for el in jit.elements:
    el.position = np.asarray(el.position, dtype=np.float64) + rng.uniform(-0.1, 0.1, 3) * np.array([1.0, 1.0, 0.0])
legs = [("tilted2_f1", tilted, focus, 200), ("tilted2_f8", tilted, shard, 100), ("jitter_f1", jit, focus, 100), ("jitter_f8", jit, shard, 60),
        ("jitter_sweep64", jit, sweep, 10), ("tilted2_sweep64", tilted, sweep, 20)]
for key, arr, foci, steps in legs:
    if want and key not in want:
        continue
    sf.plan_foci_sweep(arr, foci, C0, (nat.APOD_UNIFORM, 1.0, 0.0), origin, spacing, n, F0, RHO0, SENS, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
    for _ in range(5):
        sf.step("none")
    ctx.sync()
    ctx.profile_begin(steps)
    for _ in range(steps):
        sf.step("none")
    ctx.sync()
    ms = ctx.profile_end()
    print(f"{key:16s} {float(np.mean(ms)):8.4f} ms  {ctx.field_variant()[:70]}", flush=True)
