#!/usr/bin/env python3
"""Developer tool: timeline of ONE Protocol.calc_solution(simulate=True, scale=True) on the bench's 8-focus shard -- when every C-ABI crossing
starts and how long it blocks, and the Python time between them (medians over 20 calls)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench
import openlifu_amd as ol
from openlifu_amd import _native as nat, dist as od
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = pattern.get_targets(target)
pos = np.array([f.get_position(units="m") for f in sweep])
idx = od.plan_foci_orbits(pos, 8, centre_xy=(0.0, 0.0))[0]
foci = [sweep[int(i)] for i in idx]
proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=len(foci) * 2, pulse_train_interval=0),
                    focal_pattern=bench._ListPattern(ol, foci), sim_setup=setup)
events = []
names = ["set_elements", "bf_solve", "set_steering", "field_absorption", "field_plan", "field_launch", "field_masked_peak", "solution_analyze", "field_scale", "field_aggregate_device",
         "field_scale_aggregate", "sync"]
for nm in names:
    if not hasattr(nat.Context, nm):
        continue
    def wrap(fn, nm=nm):
        def inner(self, *a, **k):
            t0 = time.perf_counter(); r = fn(self, *a, **k); events.append((nm, t0, time.perf_counter())); return r
        return inner
    setattr(nat.Context, nm, wrap(getattr(nat.Context, nm)))
# the analysis crossing runs on a helper thread: stamp when it is started (begin returns), when the host side asks for the report
# (finish called) and when the report is there (finish returns)
_begin = nat.Context.solution_analyze_begin
def begin(self, *a, **k):
    t0 = time.perf_counter(); fin = _begin(self, *a, **k); events.append(("analyze_begin", t0, time.perf_counter()))
    def finish():
        t1 = time.perf_counter(); r = fin(); events.append(("analyze_finish(wait)", t1, time.perf_counter())); return r
    finish.abandon = fin.abandon
    return finish
nat.Context.solution_analyze_begin = begin   # (since round 5 the device work is enqueued by olx_solution_analyze_begin: no helper thread)
for _ in range(3):
    proto.calc_solution(target, arr, simulate=True, scale=True)
rows = []
for _ in range(20):
    events.clear()
    t0 = time.perf_counter(); proto.calc_solution(target, arr, simulate=True, scale=True); t1 = time.perf_counter()
    rows.append([(nm, (a - t0) * 1e3, (b - a) * 1e3) for nm, a, b in events] + [("END", (t1 - t0) * 1e3, 0.0)])
n = min(len(r) for r in rows)
print("crossing                 starts at [ms]   blocks [ms]   (python before it [ms])")
prev_end = 0.0
for k in range(n):
    nm = rows[0][k][0]
    st = float(np.median([r[k][1] for r in rows])); du = float(np.median([r[k][2] for r in rows]))
    print(f"{nm:24s} {st:10.3f} {du:13.3f} {st - prev_end:18.3f}")
    prev_end = st + du
