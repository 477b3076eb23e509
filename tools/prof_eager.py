import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))
import bench
import openlifu_amd as ol
from openlifu_amd import dist as od
from openlifu_amd.util import dataset as ds
arr, setup, target, pattern = bench.synthetic_workload(256, 0.25)
sweep = pattern.get_targets(target)
pos = np.array([f.get_position(units="m") for f in sweep])
idx = od.plan_foci_orbits(pos, 8, centre_xy=(0.0, 0.0))[0]
foci = [sweep[int(i)] for i in idx]
proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=len(foci) * 2, pulse_train_interval=0),
                    focal_pattern=bench._ListPattern(ol, foci), sim_setup=setup)
class EagerDataArray(ds.DataArray):
    def __init__(self, data, coords=None, dims=None, name=None, attrs=None):
        if isinstance(data, ds.LazyDataArray): raise TypeError("cannot defer")
        super().__init__(data, coords=coords, dims=dims, name=name, attrs=attrs)
class EagerDataset(ds.Dataset):
    def __setitem__(self, name, da):
        if isinstance(da, ds.LazyDataArray): raise ValueError("MissingDimensionsError")
        super().__setitem__(name, da)
class FakeXarray:
    DataArray, Dataset, Coordinates = EagerDataArray, EagerDataset, ds.Coordinates
ds.HAVE_XARRAY, ds._xa = True, FakeXarray
for _ in range(2):
    r = proto.calc_solution(target, arr, simulate=True, scale=True); r = None
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    r = proto.calc_solution(target, arr, simulate=True, scale=True); r = None
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
