#!/usr/bin/env python3
"""Developer tool: kernel-2 time of a few steering shapes on the 256-element array x 256^3 grid (which kernel the planner picks, ms per launch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench
import openlifu_amd as ol
from openlifu_amd import _native as nat
from openlifu_amd.engine import grid_from_coords
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
origin, spacing, n = grid_from_coords(setup.get_coords())
cases = {
    "1 focus on axis": [[0, 0, 40]],
    "1 focus off axis in x": [[5, 0, 40]],
    "1 focus off both axes": [[5, 3, 40]],
    "2 foci (+-x pair)": [[5, 0, 40], [-5, 0, 40]],
    "4 foci ring": [[5, 0, 40], [-5, 0, 40], [0, 5, 40], [0, -5, 40]],
    "8 foci ring": [[5 * np.cos(a), 5 * np.sin(a), 40] for a in np.arange(8) * np.pi / 4],
    "3 foci generic": [[2, 1, 38], [-3, 4, 41], [1, -4, 43]],
}
for name, foci in cases.items():
    f = np.asarray(foci, dtype=float) * 1e-3
    ctx.bf_solve(f, 1500.0)
    for fp8 in (False, True):
        ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY | (0 if fp8 else nat.FIELD_FP16_CORRECTION))
        for _ in range(5): ctx.field_launch()
        ms = ctx.field_time(60)
        print(f"{name:26s} fp8={int(fp8)} {np.median(ms):8.4f} ms  {ctx.field_variant()[:90]}")
