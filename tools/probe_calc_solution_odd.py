import sys, time
sys.path[:0]=['/root/repo','/root/repo/openlifu-python_amd']
import numpy as np
import openlifu_amd as ol
arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=1e5)
for name, setup in (("odd 241x241x257", ol.SimSetup(spacing=0.25)),
                    ("even 240x240x256", ol.SimSetup(spacing=0.25, x_extent=(-29.875, 29.875), y_extent=(-29.875, 29.875), z_extent=(-4, 59.75)))):
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=8, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=7, spoke_radius=5.0, target_pressure=1.0e6), sim_setup=setup)
    target = ol.Point(position=(0, 0, 40), units="mm", id="t")
    ts=[]
    for rep in range(6):
        t0=time.perf_counter(); sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=True); ts.append((time.perf_counter()-t0)*1e3)
        del sol, agg, an      # (results kept alive would be rescued to the host -- 1.5 GB over PCIe -- when the next call reuses the device buffers)
    print(name, [round(t,2) for t in ts], ol.get_engine().ctx.field_variant()[:70], [c.shape for c in setup.get_coords().values()] if False else "")
