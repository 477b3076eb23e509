#!/usr/bin/env python3
"""Developer probe (GPU box): kernel 2a (general, per pair) and the planner's lattice kernel against the fp64 C oracle on the deep grid through the element
plane where tests/test_gpu_field.py::test_single_column_kernel_wide_arrays_fuzz_against_general_kernel found the two fp32 evaluations 1.07e-5 of the volume
maximum apart (36 x 13 elements, pitch 4 x 3 voxels of 0.5 mm, 144 x 57 x 350 voxels from z = -2 mm).  Prints both errors and where 2a's largest one sits."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g; g.build()
from openlifu_amd import _native as nat
from oracle import bf_oracle as bo, c_oracle as co
F0, C, RHO, P0 = 400e3, 1500.0, 1000.0, 1e5
nax, nay, mxv, myv, h = 36, 13, 4, 3, 0.5
nx, ny, nz, z0 = 144, 57, 350, -2.0
a, b = np.meshgrid(np.arange(nax), np.arange(nay), indexing="ij")
pos = np.stack([(a.ravel() - (nax - 1) / 2) * mxv * h, (b.ravel() - (nay - 1) / 2) * myv * h, np.zeros(nax * nay)], axis=1)
foci = np.array([[0.0, 0.0, (z0 + 0.5 * nz * h) * 1e-3]])
size = np.tile([0.9 * mxv * h, 0.9 * myv * h], (nax * nay, 1))
pos_m = pos * 1e-3; area = size[:, 0] * size[:, 1] * 1e-6
ctx = nat.Context(0)
ctx.set_elements(pos_m, bo.element_rotations(np.zeros_like(pos))[:, :, 2], area)
d, ap = ctx.bf_solve(foci, C, apod_kind=nat.APOD_MAXANGLE, p0=70.0, p1=0.0)
xs = (np.arange(nx) - (nx - 1) / 2) * h * 1e-3; ys = (np.arange(ny) - (ny - 1) / 2) * h * 1e-3; zs = (z0 + np.arange(nz) * h) * 1e-3
ref = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, d[0], ap[0], F0, C, P0, dmin=0.5 * h * 1e-3))
for fam in ("auto", "general", "shared", "mfma"):
    if fam == "auto": os.environ.pop("OLX_FIELD_VARIANT", None)
    else: os.environ["OLX_FIELD_VARIANT"] = fam
    ctx.field_plan((xs[0], ys[0], zs[0]), (h * 1e-3,) * 3, (nx, ny, nz), F0, C, RHO, P0, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
    ctx.field_launch()
    p = ctx.field_fetch(0)["pmag"]
    e = np.abs(p - ref)
    i = np.unravel_index(np.argmax(e), e.shape)
    print(f"{fam:8s} {ctx.field_variant()[:60]:60s} max err / max = {e.max() / ref.max():.3e} at voxel {i} (z = {zs[i[2]] * 1e3:.2f} mm), |p| there {ref[i]:.1f} of max {ref.max():.1f}; err / local = {e[i] / ref[i]:.2e}")
    far = e[:, :, 16:].max() / ref[:, :, 16:].max()
    print(f"         planes >= 16 alone: {far:.3e} of their maximum")
