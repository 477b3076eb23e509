import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "openlifu-python_amd"))
import bench, openlifu_amd as ol
from openlifu_amd import _native as nat
from openlifu_amd.engine import grid_from_coords
arr, setup, foci = bench.synthetic_workload(256, 0.25, (16, 16), 3.0, 8, seed=0)
eng = ol.get_engine(0); ctx = eng.ctx; eng.bind(arr)
ctx.bf_solve(np.array([f.get_position(units="m") for f in foci]), 1500.0)
origin, spacing, n = grid_from_coords(setup.get_coords())
ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
N = int(sys.argv[1])
ctx.profile_begin(N)
for _ in range(N):
    ctx.field_launch()
ms = np.array(ctx.profile_end())
edges = [0, 5, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000, 40000]
for a, b in zip(edges[:-1], edges[1:]):
    if a >= N: break
    seg = ms[a:min(b, N)]
    print(f"launch {a:6d}-{min(b,N):6d}: mean {seg.mean():.4f} ms  min {seg.min():.4f}  max {seg.max():.4f}  (t = {ms[:a].sum()/1e3:.2f} s)")
