import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from openlifu_amd import _native as nat, dist as od
from oracle import bf_oracle as bo, c_oracle as co
from conftest import centred_grid, synthetic_array
F0, C, RHO, P0 = 400e3, 1500.0, 1000.0, 1e5
pos, ori, size = synthetic_array(16, 16, 3.0)
sweep = bo.wheel_targets([0, 0, 40.0], True, 63, 5.0) * 1e-3
foci = sweep[od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]]
ctx = nat.Context(0)      # (developer library: OLX_LIB_PATH=.../libolx_ab.so)
pos_m = pos * 1e-3; area = size[:, 0] * size[:, 1] * 1e-6
ctx.set_elements(pos_m, bo.element_rotations(ori)[:, :, 2], area)
d, a = ctx.bf_solve(foci, C)
for n, nz in ((128, 128), (96, 50), (256, 256)):
    xs, ys, _ = centred_grid(n, 0.5 if n < 256 else 0.25)
    hh = xs[1] - xs[0]
    zs = 5e-3 + hh * np.arange(nz)
    got = {}
    for shp in ("pair", "single"):
        os.environ["OLX_COSETP_SHAPE"] = shp
        ctx.field_plan((xs[0], ys[0], zs[0]), (hh,) * 3, (n, n, nz), F0, C, RHO, P0, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
        ctx.field_launch()
        got[shp] = [ctx.field_fetch(f) for f in range(8)]
        print(n, nz, shp, ctx.field_variant()[-70:])
    same = all(np.array_equal(got["pair"][f]["pmag"], got["single"][f]["pmag"]) and np.array_equal(got["pair"][f]["intensity"], got["single"][f]["intensity"]) for f in range(8))
    ref = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, d[0], a[0], F0, C, P0, dmin=0.5 * hh))
    print("  bit-identical:", same, " single vs oracle:", float(np.abs(got["single"][0]["pmag"] - ref).max() / ref.max()), " max diff:", max(float(np.abs(got["pair"][f]["pmag"] - got["single"][f]["pmag"]).max()) for f in range(8)))
