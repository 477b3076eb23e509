import cProfile, pstats, sys, os, time
sys.path[:0] = ['/root/repo', '/root/repo/openlifu-python_amd']
import numpy as np
import openlifu_amd as ol
import bench
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = pattern.get_targets(target)
foci = [sweep[i] for i in (0,1,2,63,3,62,4,61)]
proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=16, pulse_train_interval=0),
                    focal_pattern=bench._ListPattern(ol, foci), sim_setup=setup)
proto.calc_solution(target, arr, simulate=True, scale=True)
t=time.perf_counter(); proto.calc_solution(target, arr, simulate=True, scale=True); print("calc_solution ms", (time.perf_counter()-t)*1e3)
pr = cProfile.Profile(); pr.enable()
sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
