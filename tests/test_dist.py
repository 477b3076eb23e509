"""N > 1 path on CPU: shard planner properties + a world_size-2 gloo run of the reassembly."""
import os
import subprocess
import sys

import numpy as np
import pytest

from openlifu_amd import dist as od

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("F,world", [(1, 1), (5, 2), (8, 8), (64, 8), (3, 8), (7, 4)])
def test_plan_foci_covers_every_focus_once(F, world):
    per, blocks = od.plan_foci(F, world)
    assert per * world >= F and sum(c for _, c in blocks) == F
    seen = np.concatenate([np.arange(s, s + c) for s, c in blocks])
    assert np.array_equal(seen, np.arange(F))
    for r in range(world):
        idx = od.local_focus_indices(F, world, r)
        assert len(idx) == per and idx.max() <= F - 1
        s, c = blocks[r]
        assert np.array_equal(idx[:c], np.arange(s, s + c))
    g = np.stack([np.arange(F)[od.local_focus_indices(F, world, r)] for r in range(world)])
    assert np.array_equal(od.assemble_foci(g, F), np.arange(F))


@pytest.mark.parametrize("nx,world", [(256, 8), (256, 4), (7, 2), (41, 3), (8, 8), (5, 1)])
def test_plan_slabs_tile_the_volume(nx, world):
    per, plan = od.plan_slabs(nx, world)
    owner = np.full(nx, -1)
    for r, (begin, off, cnt) in enumerate(plan):
        assert 0 <= begin and begin + per <= nx and off + cnt <= per
        assert (owner[begin + off:begin + off + cnt] == -1).all()
        owner[begin + off:begin + off + cnt] = r
    assert (owner >= 0).all()
    vol = np.arange(2 * nx * 3, dtype=np.float32).reshape(2, nx, 3, 1)
    g = np.stack([vol[:, b:b + per] for b, _, _ in plan])
    assert np.array_equal(od.assemble_slabs(g, nx), vol)
    per, plan = od.plan_slabs(2, 3)  # more ranks than planes: trailing rank owns nothing but still computes 1 plane
    assert per == 1 and plan[2][2] == 0


def test_world_size_2_gloo_reassembly():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
