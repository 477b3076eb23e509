#!/usr/bin/env python3
"""Developer tool: where does Protocol.calc_solution(simulate=True, scale=True) spend its wall time on the bench's 8-focus shard?
cProfile (cumulative) + a per-C-ABI-call timer."""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench
import openlifu_amd as ol
from openlifu_amd import dist as od
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = pattern.get_targets(target)
pos = np.array([f.get_position(units="m") for f in sweep])
idx = od.plan_foci_orbits(pos, 8, centre_xy=(0.0, 0.0))[0]
foci = [sweep[int(i)] for i in idx]
proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=len(foci) * 2, pulse_train_interval=0),
                    focal_pattern=bench._ListPattern(ol, foci), sim_setup=setup)
for _ in range(2):
    proto.calc_solution(target, arr, simulate=True, scale=True)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); proto.calc_solution(target, arr, simulate=True, scale=True); ts.append((time.perf_counter() - t0) * 1e3)
print("calc_solution wall ms:", [round(t, 2) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    proto.calc_solution(target, arr, simulate=True, scale=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
