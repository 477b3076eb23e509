"""world_size-N gloo worker for tests/test_dist.py (CPU).  Each rank plays one GPU of the sharded
field path: it computes ITS shard with the oracle (standing in for the HIP kernels, which need a
GPU), moves the padded shard through a real all_gather, and reassembles with the product's
planning / assembly code (openlifu_amd.dist).  Checks every rank ends with the full result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "openlifu-python_amd")):
    sys.path.insert(0, p)

from openlifu_amd import dist as od  # noqa: E402
from oracle import bf_oracle as bo, field_oracle as fo  # noqa: E402


def all_gather_np(x: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(x))
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return np.stack([o.numpy() for o in outs])


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pos, size, _ = bo.gen_matrix_array(4, 4, 3.0, 0.3)
    pos_m, area = pos * 1e-3, size[:, 0] * size[:, 1] * 1e-6
    xs = np.linspace(-4e-3, 4e-3, 7); ys = np.linspace(-3e-3, 3e-3, 5); zs = 5e-3 + np.arange(6) * 1e-3
    foci = bo.wheel_targets([0, 0, 30.0], True, 4, 3.0) * 1e-3  # F = 5: not divisible by 2 -> padding exercised
    F = len(foci)
    steer = [bo.beamform(pos_m, np.zeros_like(pos_m), f, 1500.0) for f in foci]
    full = np.stack([np.abs(fo.field_on_grid(xs, ys, zs, pos_m, area, d, a, 400e3, 1500.0, 1e5)) for d, a in steer]).astype(np.float32)

    # --- foci sharding
    idx = od.local_focus_indices(F, world, rank)
    local = full[idx]                                  # what this rank's GPU would have computed
    got = od.assemble_foci(all_gather_np(local), F)
    assert got.shape == full.shape and np.array_equal(got, full), "foci reassembly mismatch"

    # --- orbit-aware foci sharding (what ShardedField.plan_foci_sweep uses): the spokes of the 4-spoke wheel pair up
    shards = od.plan_foci_orbits(foci, world)
    local = full[shards[rank]]
    got = od.assemble_foci_sharded(all_gather_np(local), shards, F)
    assert np.array_equal(got, full), "orbit-sharded reassembly mismatch"
    valid = od.shard_valid_counts(shards, F)
    mine = full[shards[rank][:valid[rank]]]                 # genuine foci only: the padded aggregate (olx_field_aggregate_counts)
    tmax = torch.from_numpy(mine.max(axis=0).copy()) if len(mine) else torch.zeros(full.shape[1:])
    tsum = torch.from_numpy((mine.astype(np.float64) ** 2).sum(axis=0)) if len(mine) else torch.zeros(full.shape[1:], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    assert np.array_equal(tmax.numpy(), full.max(axis=0))
    assert np.allclose(tsum.numpy() / F, (full.astype(np.float64) ** 2).mean(axis=0), rtol=1e-12)

    # --- slab sharding (nx = 7 planes over `world` ranks: shifted trailing slab)
    per, plan = od.plan_slabs(len(xs), world)
    begin = plan[rank][0]
    local = np.stack([np.abs(fo.field_on_grid(xs[begin:begin + per], ys, zs, pos_m, area, d, a, 400e3, 1500.0, 1e5,
                                               dmin=0.5e-3)) for d, a in steer]).astype(np.float32)
    got = od.assemble_slabs(all_gather_np(local), len(xs))
    assert got.shape == full.shape and np.array_equal(got, full), "slab reassembly mismatch"

    # --- aggregated result with sharded foci: all-reduce(max) / all-reduce(sum) of one volume
    mine = full[np.unique(idx)] if rank * od.plan_foci(F, world)[0] < F else full[:0]
    owned = od.plan_foci(F, world)[1][rank]
    mine = full[owned[0]:owned[0] + owned[1]]
    tmax = torch.from_numpy(mine.max(axis=0).copy()) if len(mine) else torch.zeros(full.shape[1:])
    tsum = torch.from_numpy(mine.sum(axis=0).copy()) if len(mine) else torch.zeros(full.shape[1:])
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    assert np.array_equal(tmax.numpy(), full.max(axis=0)) and np.allclose(tsum.numpy() / F, full.mean(axis=0), rtol=1e-6)

    # --- max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == world
    objs = [b"x" * 128 if rank == 0 else None]
    dist.broadcast_object_list(objs, src=0)            # how the RCCL unique id travels
    assert objs[0] == b"x" * 128
    dist.barrier()
    if rank == 0:
        print("DIST_OK", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
