#!/usr/bin/env python3
"""Developer tool: full-volume 256^3 parity of the headline shard in the three correction modes against the fp64 C oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from openlifu_amd import _native as nat, dist as od
from oracle import bf_oracle as bo, c_oracle as co
from conftest import centred_grid, synthetic_array
F0, C, RHO, P0 = 400e3, 1500.0, 1000.0, 1e5
pos, ori, size = synthetic_array(16, 16, 3.0)
sweep = bo.wheel_targets([0, 0, 40.0], True, 63, 5.0) * 1e-3
foci = sweep[od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]]
ctx = nat.Context(0)
pos_m = pos * 1e-3; area = size[:, 0] * size[:, 1] * 1e-6
ctx.set_elements(pos_m, bo.element_rotations(ori)[:, :, 2], area)
d, a = ctx.bf_solve(foci, C)
xs, ys, zs = centred_grid(256, 0.25)
h = (xs[1] - xs[0],) * 3
refs = {f: np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, d[f], a[f], F0, C, P0, dmin=0.5 * h[0])) for f in (0, 1, 4)}
for mode, env, flags in (("fp16 (default)", {}, 0), ("mixed", {"OLX_MIXED_CORRECTION": "1"}, 0), ("fp8 (opt-in)", {}, nat.FIELD_FP8_CORRECTION)):      # (mixed: developer library, OLX_LIB_PATH=.../libolx_ab.so)
    os.environ.pop("OLX_MIXED_CORRECTION", None)
    os.environ.update(env)
    ctx.field_plan((xs[0], ys[0], zs[0]), h, (256,) * 3, F0, C, RHO, P0, flags=nat.OUT_PMAG | nat.OUT_INTENSITY | flags)
    ctx.field_launch()
    errs = [float(np.abs(ctx.field_fetch(f)["pmag"] - refs[f]).max() / refs[f].max()) for f in (0, 1, 4)]
    print(f"{mode:18s} max error / peak per focus: {['%.2e' % e for e in errs]}   {ctx.field_variant()}")
