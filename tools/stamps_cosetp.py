#!/usr/bin/env python3
"""Developer tool: phase timeline of kernel 2g (field_cosetp_k) on the headline shard from in-kernel cycle stamps.
Build:  python openlifu-python_amd/build.py -DOLX_EXP_STAMPS --out lib/libolx_STAMPS.so ; on the GPU box:
  OLX_LIB_PATH=openlifu-python_amd/lib/libolx_STAMPS.so python tools/stamps_cosetp.py [fp8]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat, dist as od  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402

fp8 = "fp8" in sys.argv[1:]
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = np.array([f.get_position(units="m") for f in pattern.get_targets(target)])
shard = od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
ctx.bf_solve(sweep[shard], 1500.0)
origin, spacing, n = grid_from_coords(setup.get_coords())
ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY | (0 if fp8 else nat.FIELD_FP16_CORRECTION))
for _ in range(30):
    ctx.field_launch()
ctx.sync()
print(ctx.field_variant())
lib = nat.load()
buf = np.zeros((4096, 8), dtype=np.uint64)
lib.olx_exp_read_stamps_cosetp.argtypes = [ctypes.c_void_p]
assert lib.olx_exp_read_stamps_cosetp(buf.ctypes.data) == 0
ok = (buf[:, 0] > 0) & (buf[:, 6] > 0)
s = buf[ok].astype(np.int64)
names = ["start -> steering chunk staged (barrier)", "table generation, pair 0", "barrier", "K-steps, pair 0", "remaining pairs", "|p| + stores issued"]
d = np.diff(s[:, :7], axis=1)
print(f"{ok.sum()} waves sampled; shader cycles, median / p10 / p90")
for k, nm in enumerate(names):
    print(f"  {nm:45s} {np.median(d[:, k]):10.0f} {np.percentile(d[:, k], 10):10.0f} {np.percentile(d[:, k], 90):10.0f}")
if (buf[ok][:, 7] > 0).all():      # second pair in detail: wait at its barrier (drains the loads in flight), then tables (evaluated or copied) + barrier + K-steps
    w2 = s[:, 7] - s[:, 4]; r2 = s[:, 5] - s[:, 7]
    print(f"  {'  pair 1: barrier on entry (vmcnt drain)':45s} {np.median(w2):10.0f} {np.percentile(w2, 10):10.0f} {np.percentile(w2, 90):10.0f}")
    print(f"  {'  pair 1: tables + barrier + K-steps':45s} {np.median(r2):10.0f} {np.percentile(r2, 10):10.0f} {np.percentile(r2, 90):10.0f}")
tot = s[:, 6] - s[:, 0]
print(f"  {'wave lifetime':45s} {np.median(tot):10.0f} {np.percentile(tot, 10):10.0f} {np.percentile(tot, 90):10.0f}")
