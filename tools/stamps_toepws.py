#!/usr/bin/env python3
"""Developer tool: step timeline of the persistent kernel 2f (field_toepws_k) from in-kernel cycle stamps: how long the two
teams of a block take for one step (stamps around step 4: even step = first super-block of an item, odd = last + stores).
Build:  python openlifu-python_amd/build.py -DOLX_EXP_STAMPS --out lib/libolx_STAMPS.so ; on the GPU box:
  OLX_LIB_PATH=openlifu-python_amd/lib/libolx_STAMPS.so python tools/stamps_toepws.py [grid] [spacing_mm] [elements] [pitch_mm]"""
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
lib.olx_exp_read_stamps_toepws.argtypes = [ctypes.c_void_p]
assert lib.olx_exp_read_stamps_toepws(buf.ctypes.data) == 0
s = buf.astype(np.int64).reshape(512, 8, 8)
ok = s[:, 0, 0] > 0
s = s[ok]
for team, sl in (("contractors (waves 0-3)", slice(0, 4)), ("generators (waves 4-7)", slice(4, 8))):
    t = s[:, sl, :]
    print(f"{team}: {ok.sum()} blocks; shader cycles, median / p10 / p90")
    for nm, a, b in (("wait at the hand-over barrier (step 4)", 0, 1), ("own work of step 4", 1, 2), ("step 5 incl. its barrier", 2, 3)):
        d = (t[:, :, b] - t[:, :, a]).ravel()
        print(f"  {nm:42s} {np.median(d):9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")
