#!/usr/bin/env python3
"""Developer smoke: native kernels vs the oracle on a few sizes + raw kernel timing.
(Checker script, not product: it imports oracle/.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "openlifu-python_amd"))

from openlifu_amd import _native as nat  # noqa: E402
from oracle import bf_oracle as bo, c_oracle as co  # noqa: E402


def case(nx_el, ny_el, pitch, ngrid, spacing_mm, focus_mm=(0, 0, 40), jitter=False, check=True, iters=10):
    rng = np.random.default_rng(147)
    pos, size, _ = bo.gen_matrix_array(nx_el, ny_el, pitch, 0.1 * pitch)
    ori = np.zeros_like(pos)
    if jitter:
        pos = pos + rng.uniform(-0.1, 0.1, pos.shape)
        ori = np.deg2rad(rng.uniform(-5, 5, pos.shape))
    pos_m = pos * 1e-3
    area = size[:, 0] * size[:, 1] * 1e-6
    nrm = bo.element_rotations(ori)[:, :, 2]
    focus = np.array(focus_mm) * 1e-3
    c, rho, f0 = 1500.0, 1000.0, 400e3
    h = spacing_mm * 1e-3
    xs = (np.arange(ngrid) - ngrid / 2) * h
    zs = 5e-3 + np.arange(ngrid) * h
    ctx = nat.Context(0)
    ctx.set_elements(pos_m, nrm, area)
    d, a = ctx.bf_solve(focus[None, :], c, apod_kind=nat.APOD_UNIFORM, p0=1.0)
    d_o, a_o = bo.beamform(pos_m, ori, focus, c)
    print(f"[bf] N={len(pos)} max|d-d_o|/max d = {np.abs(d[0]-d_o).max()/d_o.max():.2e}  apod eq {np.array_equal(a[0], a_o)}")
    ctx.field_plan((xs[0], xs[0], zs[0]), (h, h, h), (ngrid,) * 3, f0, c, rho, 1e5,
                   flags=nat.OUT_PMAG | nat.OUT_INTENSITY | nat.OUT_COMPLEX)
    ctx.field_launch(); ctx.sync()
    ms = ctx.field_time(iters)
    pairs = ngrid ** 3 * len(pos)
    print(f"[field] {ctx.field_variant()} N={len(pos)} grid={ngrid}^3: median {np.median(ms):.4f} ms  min {ms.min():.4f} "
          f"-> {pairs/np.median(ms)/1e3/1e6:.1f} G pairs/s  ({ngrid**3*4/np.median(ms)/1e6:.1f} GB/s |p| only)")
    if check:
        out = ctx.field_fetch(0, want=("pmag", "intensity", "complex"))
        t = time.time()
        p_o = co.field_on_grid(xs, xs, zs, pos_m, area, d_o, a_o, f0, c, 1e5)
        print(f"        oracle {time.time()-t:.2f}s")
        mx = np.abs(p_o).max()
        print(f"        max| |p|-|p_o| |/max|p_o| = {np.abs(out['pmag']-np.abs(p_o)).max()/mx:.3e}   "
              f"complex: {np.abs(out['complex']-p_o).max()/mx:.3e}   "
              f"intensity: {np.abs(out['intensity']-1e-4*np.abs(p_o)**2/(2*rho*c)).max()/(1e-4*mx**2/(2*rho*c)):.3e}")
    ctx.close()


if __name__ == "__main__":
    case(8, 8, 4.0, 32, 1.0)
    case(16, 16, 3.0, 64, 0.5, jitter=True)
    case(16, 16, 3.0, 128, 0.5)
    case(16, 16, 3.0, 256, 0.25, check=False, iters=20)
    case(32, 32, 1.5, 256, 0.25, check=False, iters=5)
