"""Checker script (not a pytest module): field error of every kernel family against the fp64 oracle on an 8-focus Wheel.
Run from the repo root on the GPU box:  python tests/field_error_report.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "openlifu-python_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from openlifu_amd import _native as nat
from oracle import bf_oracle as bo, c_oracle as co
from conftest import centred_grid, synthetic_array
F0, C, RHO, P0 = 400e3, 1500.0, 1000.0, 1e5
ctx = nat.Context(0)
pos, ori, size = synthetic_array(16, 16, 3.0)
foci = bo.wheel_targets([0, 0, 40.0], True, 7, 5.0) * 1e-3
pos_m = pos * 1e-3; area = size[:, 0] * size[:, 1] * 1e-6
ctx.set_elements(pos_m, bo.element_rotations(ori)[:, :, 2], area)
st = [bo.beamform(pos_m, ori, f, C) for f in foci]
d = np.array([s[0] for s in st]); a = np.array([s[1] for s in st])
ctx.set_steering(d, a)
for n, h in ((128, 0.5), (64, 1.0)):
    xs, ys, zs = centred_grid(n, h)
    for var in ("lattice", "lattice2d", "mfma", "shared"):
        os.environ["OLX_FIELD_VARIANT"] = var
        ctx.field_plan((xs[0], ys[0], zs[0]), (xs[1] - xs[0],) * 3, (n,) * 3, F0, C, RHO, P0)
        ctx.field_launch()
        worst = 0
        for f in (0, 3):
            ref = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, d[f], a[f], F0, C, P0))
            worst = max(worst, np.abs(ctx.field_fetch(f)["pmag"] - ref).max() / ref.max())
        print(n, var, ctx.field_variant()[:28], f"{worst:.2e}")
