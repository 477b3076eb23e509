#!/usr/bin/env python3
"""Developer tool: device -> fresh NumPy array rates of the staged fetch (olx_aggregate_fetch: one 67 MB volume;
olx_field_fetch_all: 8 volumes = 537 MB) against worker count and pinned-chunk size (OLX_FETCH_THREADS, OLX_FETCH_CHUNK_KB)."""
import os, sys, time
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
ctx.field_launch(); ctx.field_aggregate(); ctx.sync()
combos = [(t, c) for c in (8192, 4096, 2048, 1024, 512) for t in (4, 8, 12, 16)]
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
print(f"{'threads':>7s} {'chunk KB':>8s} {'67 MB ms':>9s} {'GB/s':>6s} {'537 MB ms':>10s} {'GB/s':>6s}")
for thr, chunk in combos:
    os.environ["OLX_FETCH_THREADS"] = str(thr)
    os.environ.pop("OLX_FETCH_CHUNK_KB", None)         # chunk 0 = the library's own choice
    if chunk:
        os.environ["OLX_FETCH_CHUNK_KB"] = str(chunk)
    ctx.aggregate_fetch(want_intensity=False)          # (re)allocates the pinned chunks of this size
    small, big = [], []
    for _ in range(5):
        t0 = time.perf_counter(); pm, _i = ctx.aggregate_fetch(want_intensity=False); small.append(time.perf_counter() - t0); del pm
    for _ in range(3):
        t0 = time.perf_counter(); r = ctx.field_fetch_all(want=("pmag",)); big.append(time.perf_counter() - t0); del r
    s, b = float(np.median(small)), float(np.median(big))
    print(f"{thr:7d} {chunk:8d} {s * 1e3:9.2f} {67.108864e-3 / s:6.1f} {b * 1e3:10.2f} {536.870912e-3 / b:6.1f}", flush=True)
