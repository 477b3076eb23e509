"""Multi-GPU sharding of the field path: one process per GPU, foci or x-slabs per rank, results
reassembled with an RCCL all-gather over xGMI.

The path has no exchange step for correctness -- foci and voxels are independent.  The collective
exists only because the API returns every per-focus volume to the caller
(simulation_result[focal_point_index, x, y, z], plan/protocol.py:341-347).

Planning / assembly are pure functions (covered on CPU by the world_size-2 gloo test); ``ShardedField``
drives them on GPUs through the C-ABI (olx_comm_init / olx_field_allgather).  RCCL all-gather needs
equal counts per rank, so shards are padded to ceil(units / world): trailing foci are repeated, trailing
slabs are shifted inwards, and ``assemble_*`` drops the padding.
"""
from __future__ import annotations

import numpy as np


# ---- planning (pure) ------------------------------------------------------------------------------
def plan_foci(n_foci: int, world: int):
    """Contiguous blocks of ``per = ceil(F / world)`` foci.  Returns (per, [(start, count)] per rank);
    count may be smaller (or 0) on trailing ranks."""
    if n_foci < 1 or world < 1:
        raise ValueError("n_foci and world must be >= 1")
    per = -(-n_foci // world)
    return per, [(min(r * per, n_foci), max(0, min(per, n_foci - r * per))) for r in range(world)]


def local_focus_indices(n_foci: int, world: int, rank: int) -> np.ndarray:
    """Indices (length ``per``) of the foci rank computes; padding repeats the last valid focus."""
    per, blocks = plan_foci(n_foci, world)
    start, count = blocks[rank]
    idx = np.arange(start, start + per)
    return np.minimum(idx, n_foci - 1)


def assemble_foci(gathered: np.ndarray, n_foci: int) -> np.ndarray:
    """gathered [world, per, ...] in rank order -> [F, ...] (padding dropped)."""
    world, per = gathered.shape[:2]
    return gathered.reshape((world * per,) + gathered.shape[2:])[:n_foci]


def plan_slabs(nx: int, world: int):
    """x-slabs (x is the slowest axis of the C-order [nx,ny,nz] volume, so a slab is one contiguous
    block).  Every rank computes exactly ``per = ceil(nx / world)`` planes; a slab that would overrun
    the grid is shifted inwards.  Returns (per, [(x_begin, valid_offset, valid_count)] per rank) where
    planes [valid_offset, valid_offset + valid_count) of the rank's slab are the ones it owns."""
    if nx < 1 or world < 1:
        raise ValueError("nx and world must be >= 1")
    per = -(-nx // world)
    out = []
    for r in range(world):
        own_lo, own_hi = min(r * per, nx), min((r + 1) * per, nx)
        begin = min(r * per, nx - per)
        out.append((begin, own_lo - begin if own_hi > own_lo else 0, own_hi - own_lo))
    return per, out


def assemble_slabs(gathered: np.ndarray, nx: int) -> np.ndarray:
    """gathered [world, F, per, ny, nz] -> [F, nx, ny, nz]."""
    world, F, per = gathered.shape[:3]
    _, plan = plan_slabs(nx, world)
    out = np.empty((F, nx) + gathered.shape[3:], dtype=gathered.dtype)
    for r, (begin, off, cnt) in enumerate(plan):
        if cnt:
            out[:, begin + off:begin + off + cnt] = gathered[r, :, off:off + cnt]
    return out


# ---- GPU driver ------------------------------------------------------------------------------------
class ShardedField:
    """Field of F foci on ``world`` GPUs.  ``exchange_id(bytes|None) -> bytes`` must broadcast rank 0's
    128-byte RCCL unique id to all ranks (e.g. via torch.distributed.broadcast_object_list over gloo,
    or an MPI / file store): the only thing the launcher has to provide."""

    def __init__(self, engine, world: int, rank: int, exchange_id):
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        ctx = engine.ctx
        if world > 1:
            uid = exchange_id(ctx.comm_unique_id() if rank == 0 else None)
            ctx.comm_init(uid, world, rank)

    def close(self):
        if self.world > 1:
            self.engine.ctx.comm_destroy()

    def sweep_foci(self, arr, foci_m, c, apod_args, origin_m, spacing_m, n, freq, rho, p0_pa):
        """mode "foci": every rank solves + accumulates its block of foci over the whole grid; returns
        |p| [F, nx, ny, nz] on every rank."""
        from . import _native as nat
        foci_m = np.atleast_2d(np.asarray(foci_m, dtype=np.float64))
        F = foci_m.shape[0]
        idx = local_focus_indices(F, self.world, self.rank)
        eng, ctx = self.engine, self.engine.ctx
        eng.bind(arr)
        kind, p0, p1 = apod_args
        ctx.bf_solve(foci_m[idx], c, apod_kind=kind, p0=p0, p1=p1, want_outputs=False)
        ctx.field_plan(origin_m, spacing_m, n, freq, c, rho, p0_pa, flags=nat.OUT_PMAG)
        ctx.field_launch()
        if self.world == 1:
            return np.stack([ctx.field_fetch(f, want=("pmag",))["pmag"] for f in range(F)])
        ctx.field_allgather()
        gathered = np.stack([ctx.allgather_fetch(r) for r in range(self.world)])
        return assemble_foci(gathered, F)

    def aggregate(self):
        """Cross-rank aggregate of the volumes the last sweep left resident: (max_f |p|, mean_f I) over ALL
        ranks' foci (plan/protocol.py:382-387) -- one RCCL all-reduce per volume instead of a gather."""
        ctx = self.engine.ctx
        if self.world == 1:
            return ctx.field_aggregate(want_intensity=bool(ctx._flags & 2))
        ctx.field_allreduce_aggregate()
        return ctx.aggregate_fetch(want_intensity=bool(ctx._flags & 2))

    def sweep_slabs(self, arr, delays, apod, origin_m, spacing_m, n, freq, c, rho, p0_pa):
        """mode "slabs": every rank accumulates ALL foci over its x-slab (better balance when
        F < world); returns |p| [F, nx, ny, nz] on every rank."""
        from . import _native as nat
        eng, ctx = self.engine, self.engine.ctx
        eng.bind(arr)
        ctx.set_steering(delays, apod)
        per, plan = plan_slabs(int(n[0]), self.world)
        ctx.field_plan(origin_m, spacing_m, n, freq, c, rho, p0_pa, flags=nat.OUT_PMAG, slab=(plan[self.rank][0], per))
        ctx.field_launch()
        F = ctx.n_foci
        if self.world == 1:
            return np.stack([ctx.field_fetch(f, want=("pmag",))["pmag"] for f in range(F)])
        ctx.field_allgather()
        gathered = np.stack([ctx.allgather_fetch(r) for r in range(self.world)])
        return assemble_slabs(gathered, int(n[0]))
