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


def test_orbit_aware_plan_keeps_mirror_partners_together():
    """BASELINE configs[2]: Wheel(center, 63 spokes) = 64 foci on 8 GPUs.  Spokes i and 63 - i are mirror images about the
    x axis (bf/focal_patterns/wheel.py:53-64): the planner hands rank 0 the centre, spoke 0 and three partner pairs, every
    other rank four pairs -- the 15 / 16 distinct steering columns per GPU the headline kernel shape is planned for."""
    from oracle import bf_oracle as bo
    sweep = bo.wheel_targets([0, 0, 40.0], True, 63, 5.0) * 1e-3
    orbits = od.mirror_orbits(sweep)
    assert len(orbits) == 33 and orbits[0] == [0] and orbits[1] == [1] and all(len(o) == 2 for o in orbits[2:])
    for a, b in orbits[2:]:
        assert np.allclose(sweep[a] * [1, -1, 1], sweep[b], atol=1e-12)
    shards = od.plan_foci_orbits(sweep, 8)
    assert [len(s) for s in shards] == [8] * 8 and od.shard_valid_counts(shards, 64) == [8] * 8
    assert list(shards[0]) == [0, 1, 2, 63, 3, 62, 4, 61]
    assert sorted(np.concatenate(shards)) == list(range(64))
    for s in shards[1:]:
        assert all(abs(sweep[s[2 * k], 1] + sweep[s[2 * k + 1], 1]) < 1e-12 for k in range(4))
    # weak scaling: an N-GPU run over the first N shards re-plans to the same shards
    for nproc in (1, 2, 4):
        sub = np.concatenate(shards[:nproc])
        again = od.plan_foci_orbits(sweep[sub], nproc)
        assert all(np.array_equal(sub[again[r]], shards[r]) for r in range(nproc))
    # an off-centre grid has no mirror planes through the pattern: every focus is its own orbit, blocks stay contiguous
    shifted = od.plan_foci_orbits(sweep, 8, centre_xy=(1.3e-3, -0.7e-3))
    assert np.array_equal(np.concatenate(shifted), np.arange(64))


@pytest.mark.parametrize("F,world", [(5, 2), (3, 8), (7, 4), (64, 8), (9, 4)])
def test_orbit_plan_padding_and_assembly(F, world):
    rng = np.random.default_rng(147)
    foci = np.column_stack([rng.uniform(-5e-3, 5e-3, F), rng.uniform(-5e-3, 5e-3, F), np.full(F, 30e-3)])
    foci[1::2, :2] = foci[0:2 * (F // 2):2, :2] * [-1, 1]          # mirror partners about the y-z plane
    shards = od.plan_foci_orbits(foci, world)
    per = -(-F // world)
    valid = od.shard_valid_counts(shards, F)
    assert all(len(s) == per for s in shards) and sum(valid) == F
    genuine = np.concatenate([s[:v] for s, v in zip(shards, valid)])
    assert sorted(genuine) == list(range(F))
    for s, v in zip(shards, valid):                                  # padding repeats an index the rank (or an earlier one) owns
        assert all(int(i) in set(genuine) for i in s[v:])
    vols = np.arange(F * 6, dtype=np.float32).reshape(F, 2, 3)
    g = np.stack([vols[s] for s in shards])
    assert np.array_equal(od.assemble_foci_sharded(g, shards, F), vols)


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


def test_bench_launcher_starts_ranks_and_relays_failure():
    """`python bench.py --gpus 2` without a launcher around it becomes the launcher: it starts the ranks itself (fresh children with
    RANK / WORLD_SIZE / MASTER_* set) and returns their worst exit code instead of hanging.  Here (no GPU) every rank fails when it
    asks for its device: the parent must come back non-zero, promptly, with nothing on stdout."""
    import subprocess
    import time
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    from openlifu_amd import _native
    if _native.device_count() > 0:
        pytest.skip("a GPU is visible: the launcher is exercised for real by tests/test_gpu_p2p.py")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-seconds", "0"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "bench launcher: rank exit codes" in p.stderr
    assert time.time() - t0 < 240
