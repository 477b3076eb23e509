#!/usr/bin/env python3
"""Developer experiment: is the launch time of the headline shard a property of the PROCESS or of the output allocation?
Plans the shard, times launches, forces the output volumes to be re-allocated (a larger plan, then the shard again) and times again.
  python tools/exp_store_modes.py [rounds]      (OLX_LIB_PATH selects the build, e.g. the store-only timing build)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat, dist as od  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = np.array([f.get_position(units="m") for f in pattern.get_targets(target)])
shard = od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
origin, spacing, n = grid_from_coords(setup.get_coords())
for r in range(rounds):
    ctx.bf_solve(sweep[shard], 1500.0)
    ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
    for _ in range(30):
        ctx.field_launch()
    ctx.sync()
    t = np.asarray(ctx.field_time(100))
    print(f"round {r}: kernel {np.mean(t):.4f} ms (min {np.min(t):.4f}, max {np.max(t):.4f})  {ctx.field_variant()[:40]}", flush=True)
    # re-allocation: a plan with more foci frees and re-allocates the volumes; then back
    ctx.bf_solve(sweep[:9 + r], 1500.0)
    ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
    ctx.field_launch(); ctx.sync()
