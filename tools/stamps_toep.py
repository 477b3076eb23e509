#!/usr/bin/env python3
"""Developer tool: phase timeline of kernel 2f (field_toep_k) from in-kernel cycle stamps.
Build:  python openlifu-python_amd/build.py -DOLX_EXP_STAMPS --out lib/libolx_STAMPS.so ; on the GPU box:
  OLX_LIB_PATH=openlifu-python_amd/lib/libolx_STAMPS.so python tools/stamps_toep.py [grid] [spacing_mm] [elements] [pitch_mm]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
el = tuple(int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "16x16").split("x"))
pitch = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
arr, setup, target, pattern = bench.synthetic_workload(grid, sp, el, pitch)
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
ctx.bf_solve(np.array([target.get_position(units="m")]), 1500.0)
origin, spacing, n = grid_from_coords(setup.get_coords())
ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
for _ in range(30):
    ctx.field_launch()
ctx.sync()
print(ctx.field_variant())
lib = nat.load()
buf = np.zeros((4096, 8), dtype=np.uint64)
lib.olx_exp_read_stamps_toep.argtypes = [ctypes.c_void_p]
assert lib.olx_exp_read_stamps_toep(buf.ctypes.data) == 0
ok = buf[:, 0] > 0
s = buf[ok].astype(np.int64)
names = ["start (zero fill, A loads issued) -> table free", "table generation, super-block 0", "barrier", "contraction, super-block 0",
         "remaining super-blocks", "epilogue (|p|, stores issued)"]
d = np.diff(s[:, :7], axis=1)
print(f"{ok.sum()} waves sampled; shader cycles, median / p10 / p90")
for k, nm in enumerate(names):
    print(f"  {nm:50s} {np.median(d[:, k]):10.0f} {np.percentile(d[:, k], 10):10.0f} {np.percentile(d[:, k], 90):10.0f}")
tot = s[:, 6] - s[:, 0]
print(f"  {'wave lifetime':50s} {np.median(tot):10.0f} {np.percentile(tot, 10):10.0f} {np.percentile(tot, 90):10.0f}")
print(f"  kernel span (first start -> last end): {s[:, 6].max() - s[:, 0].min()}")
