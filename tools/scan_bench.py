#!/usr/bin/env python3
"""Developer tool: GB/s of the HBM-bound streaming scans (olx_scan_time) over the resident result of the headline shard."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench
import openlifu_amd as ol
from openlifu_amd import _native as nat, dist as od
from openlifu_amd.engine import grid_from_coords
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = np.array([f.get_position(units="m") for f in pattern.get_targets(target)])
origin, spacing, n = grid_from_coords(setup.get_coords())
eng = ol.get_engine(0)
sf = od.ShardedField(eng, 1, 0)
sf.plan_foci_sweep(arr, sweep[od.plan_foci_orbits(sweep, 8)[0]], 1500.0, (nat.APOD_UNIFORM, 1.0, 0.0), origin, spacing, n, 400e3, 1000.0, 1e5,
                   flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
ctx = eng.ctx
for _ in range(50):
    ctx.field_launch()
ctx.sync()
for k in ("aggregate", "scale", "analysis_peaks", "masked_peak", "weighted_sum", "offset_grid", "fused_post"):
    ctx.scan_time(k, 10)
    ms, nb = ctx.scan_time(k, 50)
    t = float(np.mean(ms))
    print(f"{k:16s} {t * 1e3:8.1f} us  {nb / 1e6:8.1f} MB  {nb / t / 1e6:8.1f} GB/s  {nb / t / 1e6 / 8000 * 100:5.1f} % of 8 TB/s")
