#!/usr/bin/env python3
"""Developer tool: where does an A/B form of kernel 2g (OLX_FIELD_VARIANT=...) differ from the default launch?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import openlifu_amd as ol
from openlifu_amd import _native as nat, dist as od
from openlifu_amd.engine import grid_from_coords
fam = sys.argv[1]; fp8 = "fp8" in sys.argv[2:]
n = 128
arr, setup, target, pattern = bench.synthetic_workload(n, 0.5)
sweep = np.array([f.get_position(units="m") for f in pattern.get_targets(target)])
shard = od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
ctx.bf_solve(sweep[shard], 1500.0)
origin, spacing, nn = grid_from_coords(setup.get_coords())
got = {}
for f in (None, fam):
    if f is None: os.environ.pop("OLX_FIELD_VARIANT", None)
    else: os.environ["OLX_FIELD_VARIANT"] = f
    ctx.field_plan(origin, spacing, nn, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY | (nat.FIELD_FP8_CORRECTION if fp8 else 0))
    ctx.field_launch(); ctx.sync()
    print(ctx.field_variant())
    got[f] = [ctx.field_fetch(k)["pmag"].copy() for k in range(len(shard))]
for k in range(len(shard)):
    d = got[None][k] != got[fam][k]
    idx = np.argwhere(d)
    print("focus", k, "differing voxels", idx.shape[0])
    if idx.shape[0]:
        print("  i range", idx[:, 0].min(), idx[:, 0].max(), " j range", idx[:, 1].min(), idx[:, 1].max(), " k range", idx[:, 2].min(), idx[:, 2].max())
        print("  k mod 16 histogram", np.bincount(idx[:, 2] % 16, minlength=16))
        print("  k // 16 histogram", np.bincount(idx[:, 2] // 16, minlength=8))
        print("  first", idx[:5].tolist(), "values", [(float(got[None][k][tuple(q)]), float(got[fam][k][tuple(q)])) for q in idx[:5]])
