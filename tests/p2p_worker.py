"""One rank of tests/test_gpu_p2p.py: a FRESH process (started by subprocess before anything in it touched the GPU) that shares
device 0 with its peer rank and exchanges its shards through the product's peer-to-peer transport (OLX_GATHER=p2p: HIP IPC +
a shared-memory control block).  Rendezvous between the ranks = files in a scratch directory."""
import os
import sys
import time

import numpy as np

rank, world, tmp, n_foci = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "openlifu-python_amd")):
    sys.path.insert(0, p)
os.environ["OLX_GATHER"] = "p2p"
os.environ.setdefault("OLX_P2P_TIMEOUT_S", "60")

import openlifu_amd as ol                      # noqa: E402
from openlifu_amd import _native as nat, dist as od   # noqa: E402

_round = [0]


def allgather_bytes(blob: bytes):
    k = _round[0]
    _round[0] += 1
    mine = os.path.join(tmp, f"x{k}_{rank}")
    with open(mine + ".tmp", "wb") as f:
        f.write(blob)
    os.rename(mine + ".tmp", mine)
    out, t0 = [], time.time()
    for r in range(world):
        path = os.path.join(tmp, f"x{k}_{r}")
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise TimeoutError(f"rank {rank}: no file from rank {r} in round {k}")
            time.sleep(0.002)
        with open(path, "rb") as f:
            out.append(f.read())
    return out


def exchange_id(uid):
    return allgather_bytes(uid or b"")[0]


F0, C, RHO, P0 = 400e3, 1500.0, 1000.0, 1e5
arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=None)
n = (45, 40, 48)
spacing = (0.5e-3,) * 3
origin = (-(n[0] - 1) / 2 * spacing[0], -(n[1] - 1) / 2 * spacing[1], 5e-3)
rng = np.random.default_rng(147)
foci = np.column_stack([rng.uniform(-3e-3, 3e-3, n_foci), rng.uniform(-3e-3, 3e-3, n_foci), rng.uniform(15e-3, 25e-3, n_foci)])
foci[0] = [0, 0, 20e-3]
eng = ol.get_engine(0)
sf = od.ShardedField(eng, world, rank)
sf.init_comm(exchange_id, allgather_bytes)
assert sf.transport == "p2p" and eng.ctx.comm_transport() == "p2p", sf.transport
res = {}
# foci shards (F not divisible by the world: the last shard is padded), one step, then three more steps through both output buffers
sf.plan_foci_sweep(arr, foci, C, (nat.APOD_UNIFORM, 1.0, 0.0), origin, spacing, n, F0, RHO, P0, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
sf.step("allgather")
res["foci"] = sf.fetch_all()
for _ in range(3):
    sf.step("allgather")
res["foci_again"] = sf.fetch_all()
# launches that are NOT gathered, a different number on every rank (a time-based clock ramp does this): the same generation now sits in
# different output buffers on the two ranks, and the exchange must not care
for _ in range(rank + 1):
    eng.ctx.field_launch()
eng.ctx.sync()
for _ in range(3):
    sf.step("allgather")
res["foci_skewed"] = sf.fetch_all()
# the aggregate over ALL ranks' genuine foci through the same transport: all-reduce (every rank holds the whole volume pair), then the
# reduce-scatter form (rank r owns voxels [r V / N, (r + 1) V / N)), twice each so that the buffers are re-used while peers may still pull
for _ in range(2):
    pm, im = sf.aggregate()
res["agg_p"], res["agg_i"] = pm, im
for _ in range(2):
    eng.ctx.field_reduce_scatter_aggregate()
    pm, im = eng.ctx.aggregate_fetch()
res["rs_p"], res["rs_i"] = pm, im
# x-slabs (nx odd: the last slab is shifted inwards), all foci on every rank
d, a = eng.beamform(arr, foci, C)
res["slabs"] = sf.sweep_slabs(arr, d, a, origin, spacing, n, F0, C, RHO, P0)
sf.step("allgather"); sf.step("allgather")
res["slabs_again"] = sf.fetch_all()
# a rank that re-plans to a LARGER grid right after its own fetch, while a slower peer has not pulled that step yet: the plan frees
# the exported blocks, so it has to wait for the peers' pulls first (use-after-free of an IPC mapping otherwise)
time.sleep(0.5 * rank)
sf.step("allgather")
res["slabs_skewed"] = sf.fetch_all()
n2 = (n[0] + 4, n[1] + 4, n[2] + 4)
origin2 = (-(n2[0] - 1) / 2 * spacing[0], -(n2[1] - 1) / 2 * spacing[1], 5e-3)
res["big"] = sf.sweep_slabs(arr, d, a, origin2, spacing, n2, F0, C, RHO, P0)
# ... and a rank that scales its volumes in place while the peer is late with its pull: no half-scaled block may travel
time.sleep(0.3 * (world - 1 - rank))
sf.step("allgather")
eng.ctx.field_scale(np.full(eng.ctx.n_foci, 2.0))
res["big_after_scale"] = sf.fetch_all()
# BASELINE configs[4]'s split: a heterogeneous medium (skull slab, marched ray sums), x-slabs per rank, the medium volumes replicated --
# every rank marches the running ray sums over the WHOLE lateral grid and evaluates its slab; two steps through both output buffers
from openlifu_amd.seg.seg_methods import skull_slab_volumes   # noqa: E402
axes = [origin[a] + np.arange(n[a]) * spacing[a] for a in range(3)]
skull = skull_slab_volumes(*axes)
skull["model"] = "marched"
res["hetero_slabs"] = sf.sweep_slabs(arr, d, a, origin, spacing, n, F0, C, RHO, P0, medium=skull)
assert "field_hmarch_k" in eng.ctx.field_variant(), eng.ctx.field_variant()
sf.step("allgather")
res["hetero_slabs_again"] = sf.fetch_all()
np.savez(os.path.join(tmp, f"out_{rank}.npz"), **res)
sf.close()
print(f"rank {rank}: ok", flush=True)
