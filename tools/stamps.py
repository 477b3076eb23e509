#!/usr/bin/env python3
"""Developer tool: phase timeline of the lattice kernels from in-kernel cycle stamps (one wave in 37 blocks).
Build:  hipcc ... -DOLX_EXP_STAMPS -o openlifu-python_amd/lib/libolx_STAMPS.so ; run on the GPU box:
  OLX_LIB_PATH=.../libolx_STAMPS.so python tools/stamps.py [foci] [2d]        (default: kernel 2e, "2d": kernel 2d)
"""
import ctypes
import os
import sys

K2D = "2d" in sys.argv[2:]
if K2D:
    os.environ.setdefault("OLX_FIELD_VARIANT", "lattice2d")

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench  # noqa: E402
import openlifu_amd as ol  # noqa: E402
from openlifu_amd import _native as nat  # noqa: E402
from openlifu_amd.engine import grid_from_coords  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arr, setup, foci = bench.synthetic_workload(256, 0.25, (16, 16), 3.0, F, seed=0)
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
ctx.bf_solve(np.array([f.get_position(units="m") for f in foci]), 1500.0)
origin, spacing, n = grid_from_coords(setup.get_coords())
ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
for _ in range(3):
    ctx.field_launch()
ctx.sync()
print(ctx.field_variant())
lib = nat.load()
buf = np.zeros((4096, 8), dtype=np.uint64)
lib.olx_exp_read_stamps.argtypes = [ctypes.c_void_p]
assert lib.olx_exp_read_stamps(buf.ctypes.data) == 0
ok = buf[:, 0] > 0
s = buf[ok].astype(np.int64)
if K2D:
    names = ["start->B staged", "t-gen sb0", "4 K-steps sb0", "rest of K loop", "epilogue stage+sync", "read-out stores"]
    last = 6
else:
    names = ["start->B staged", "t-gen sb0", "4 K-steps sb0", "rest of K loop", "wait for the block (barrier)",
             "|p| + staging + barrier", "read-out + store issue"]
    last = 7
d = np.diff(s[:, :last + 1], axis=1)
print(f"{ok.sum()} waves sampled; counter ticks (s_memtime), median / p10 / p90")
for k, nm in enumerate(names):
    print(f"  {nm:24s} {np.median(d[:, k]):10.0f} {np.percentile(d[:, k], 10):10.0f} {np.percentile(d[:, k], 90):10.0f}")
tot = s[:, last] - s[:, 0]
print(f"  {'wave lifetime':24s} {np.median(tot):10.0f} {np.percentile(tot, 10):10.0f} {np.percentile(tot, 90):10.0f}")
print(f"  kernel span (first start -> last end): {s[:, last].max() - s[:, 0].min()}")
