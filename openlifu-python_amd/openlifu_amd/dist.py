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


def mirror_orbits(foci_m, centre_xy=(0.0, 0.0), tol: float = 1e-9):
    """Groups of foci that are mirror images of one another about the planes x = cx and y = cy (the symmetry planes
    of a grid centred on a symmetric array), in order of first appearance.  The steering vectors of an orbit coincide
    up to the array's mirror permutations, so a GPU that holds a whole orbit accumulates each distinct vector once
    (kernel 2c/2e column plan): the mirror-partner spokes (i, n - i) of a ``Wheel``
    (bf/focal_patterns/wheel.py:53-64) are each other's images."""
    foci_m = np.atleast_2d(np.asarray(foci_m, dtype=np.float64))
    keys, orbits = {}, []
    for i, f in enumerate(foci_m):
        k = (int(round(abs(f[0] - centre_xy[0]) / tol)), int(round(abs(f[1] - centre_xy[1]) / tol)), int(round(f[2] / tol)))
        if k not in keys:
            keys[k] = len(orbits)
            orbits.append([])
        orbits[keys[k]].append(i)
    return orbits


def plan_foci_orbits(foci_m, world: int, centre_xy=(0.0, 0.0), tol: float = 1e-9):
    """Orbit-aware shard plan: ``per = ceil(F / world)`` focus indices per rank (a list of ``world`` int arrays of
    length ``per``), whole mirror orbits kept on one rank whenever they fit (first-fit over the orbits in order of
    first appearance; an orbit is split only when no remaining orbit fills the gap).  Ranks past the last focus and
    the tail of the last used rank repeat that rank's last valid focus (equal counts for the RCCL all-gather);
    ``shard_valid_counts`` gives the number of genuine entries."""
    foci_m = np.atleast_2d(np.asarray(foci_m, dtype=np.float64))
    F = foci_m.shape[0]
    if F < 1 or world < 1:
        raise ValueError("need at least one focus and one rank")
    per = -(-F // world)
    todo = [list(o) for o in mirror_orbits(foci_m, centre_xy, tol)]
    shards = []
    for _ in range(world):
        mine = []
        while len(mine) < per and todo:
            room = per - len(mine)
            fit = next((q for q, o in enumerate(todo) if len(o) <= room), None)
            if fit is None:                      # nothing fits whole: split the first orbit
                mine += todo[0][:room]
                todo[0] = todo[0][room:]
            else:
                mine += todo.pop(fit)
        shards.append(mine)
    out = []
    for mine in shards:
        pad = mine[-1] if mine else F - 1
        out.append(np.array(mine + [pad] * (per - len(mine)), dtype=np.int64))
    return out


def shard_valid_counts(shards, n_foci: int):
    """Genuine (non-padding) entries per rank of a ``plan_foci_orbits`` plan: every focus counted exactly once."""
    seen, counts = set(), []
    for sh in shards:
        cnt = 0
        for i in sh:                 # genuine entries come first; padding repeats an index already counted
            if int(i) in seen:
                break
            seen.add(int(i)); cnt += 1
        counts.append(cnt)
    assert len(seen) == n_foci, "plan does not cover every focus"
    return counts


def assemble_foci_sharded(gathered: np.ndarray, shards, n_foci: int) -> np.ndarray:
    """gathered [world, per, ...] in rank order -> [F, ...] in the caller's focus order."""
    out = np.empty((n_foci,) + gathered.shape[2:], dtype=gathered.dtype)
    for r, sh in enumerate(shards):
        out[np.asarray(sh)] = gathered[r]
    return out


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
    or an MPI / file store): the only thing the launcher has to provide.

    ``plan_foci_sweep`` / ``plan_slab_sweep`` leave everything resident on the device; ``step`` is ONE pass of the
    hot path (launch + the chosen exchange, asynchronous); ``fetch_*`` brings results to the host.  bench.py times
    ``step``; ``sweep_foci`` / ``sweep_slabs`` are plan + step + fetch."""

    def __init__(self, engine, world: int, rank: int, exchange_id=None, allgather_bytes=None):
        self.engine, self.world, self.rank = engine, int(world), int(rank)
        self.shards, self.F, self.mode, self.comm = None, 0, None, False
        self.transport, self._allgather_bytes = "", None
        if world > 1 and exchange_id is not None:
            self.init_comm(exchange_id, allgather_bytes)

    def init_comm(self, exchange_id, allgather_bytes=None):
        """``exchange_id`` broadcasts rank 0's 128-byte id.  Which transport it names is rank 0's choice (environment
        ``OLX_GATHER=rccl|p2p``, include/olx.h); the direct peer-to-peer transport also needs ``allgather_bytes(blob) ->
        [blob of rank 0, blob of rank 1, ...]`` from the launcher: after every plan the ranks exchange the IPC handles of their
        output blocks through it."""
        ctx = self.engine.ctx
        uid = exchange_id(ctx.comm_unique_id() if self.rank == 0 else None)
        ctx.comm_init(uid, self.world, self.rank)
        self.comm = True
        self.transport = ctx.comm_transport()
        self._allgather_bytes = allgather_bytes
        if self.transport == "p2p" and allgather_bytes is None:
            raise ValueError("the p2p transport needs allgather_bytes (exchange of the ranks' IPC exports after every plan)")

    def _publish_blocks(self):
        """p2p transport: the output blocks were (re)allocated by the plan -- exchange their IPC handles."""
        if self.comm and self.transport == "p2p":
            ctx = self.engine.ctx
            ctx.comm_import(self._allgather_bytes(ctx.comm_export()))

    def close(self):
        if self.comm:
            self.engine.ctx.comm_destroy()
            self.comm = False

    # ---- planning ---------------------------------------------------------------------------------
    def plan_foci_sweep(self, arr, foci_m, c, apod_args, origin_m, spacing_m, n, freq, rho, p0_pa, flags=None,
                        fp8_correction=None, absorption=0.0):
        """mode "foci": this rank solves (kernel 1) and plans its orbit-aware block of foci over the whole grid."""
        from . import _native as nat
        foci_m = np.atleast_2d(np.asarray(foci_m, dtype=np.float64))
        self.F, self.mode = foci_m.shape[0], "foci"
        centre = tuple(origin_m[a] + 0.5 * (int(n[a]) - 1) * spacing_m[a] for a in (0, 1))
        self.shards = plan_foci_orbits(foci_m, self.world, centre_xy=centre, tol=1e-9)
        self.valid = shard_valid_counts(self.shards, self.F)
        eng, ctx = self.engine, self.engine.ctx
        eng.retire_results()
        eng.bind(arr)
        kind, p0, p1 = apod_args
        ctx.bf_solve(foci_m[self.shards[self.rank]], c, apod_kind=kind, p0=p0, p1=p1, want_outputs=False)
        flags = nat.OUT_PMAG if flags is None else flags
        ctx.field_absorption(absorption)       # (sticky context state: always say what this plan means)
        ctx.field_plan(origin_m, spacing_m, n, freq, c, rho, p0_pa,
                       flags=flags | (nat.FIELD_FP16_CORRECTION if fp8_correction is False else 0))
        ctx.aggregate_counts(self.valid[self.rank], self.F)
        eng.result_token += 1
        self._publish_blocks()
        return self.shards[self.rank]

    def plan_slab_sweep(self, arr, delays, apod, origin_m, spacing_m, n, freq, c, rho, p0_pa, flags=None, medium=None, absorption=0.0):
        """mode "slabs": every rank accumulates ALL foci over its x-slab (better balance when F < world; what the
        heterogeneous configuration uses -- ``medium`` = dict of WHOLE-grid volumes, replicated on every rank because
        the rays to a slab cross the full lateral extent)."""
        from . import _native as nat
        eng, ctx = self.engine, self.engine.ctx
        eng.retire_results()
        eng.bind(arr)
        ctx.set_steering(delays, apod)
        self.F, self.mode = ctx.n_foci, "slabs"
        self.nx = int(n[0])
        per, plan = plan_slabs(self.nx, self.world)
        self.slab = (plan[self.rank][0], per)
        ctx.field_absorption(0.0 if medium is not None else absorption)
        ctx.field_plan(origin_m, spacing_m, n, freq, c, rho, p0_pa, flags=nat.OUT_PMAG if flags is None else flags,
                       slab=self.slab)
        if medium is not None:
            ctx.field_set_medium(medium.get("sound_speed"), medium.get("attenuation"), medium.get("density"),
                                 planes_per_layer=int(medium.get("planes_per_layer", 1)), model=medium.get("model", "auto"))
        eng.result_token += 1
        self._publish_blocks()
        return self.slab

    # ---- one pass ---------------------------------------------------------------------------------
    def step(self, reassemble="allgather"):
        """Launch the planned accumulate and, with a communicator, enqueue the exchange on the side stream:
        "allgather" (every per-focus |p| volume to every rank -- what the API returns, plan/protocol.py:341-347),
        "aggregate" (reduce-scatter of max |p| / mean intensity, protocol.py:382-387) or "none"."""
        ctx = self.engine.ctx
        ctx.field_launch()
        if not self.comm or reassemble == "none":
            return
        if reassemble == "allgather":
            ctx.field_allgather()
        elif reassemble == "aggregate":
            ctx.field_reduce_scatter_aggregate()
        else:
            raise ValueError(f"unknown reassembly {reassemble!r}")

    # ---- results ----------------------------------------------------------------------------------
    def fetch_all(self):
        """|p| [F, nx, ny, nz] on this rank after ``step("allgather")`` (or the local volumes when world == 1)."""
        ctx = self.engine.ctx
        if not self.comm:
            local = np.stack([ctx.field_fetch(f, want=("pmag",))["pmag"] for f in range(ctx.n_foci)])
            gathered = local[None]
        else:
            gathered = np.stack([ctx.allgather_fetch(r) for r in range(self.world)])
        if self.mode == "foci":
            return assemble_foci_sharded(gathered, self.shards, self.F)
        return assemble_slabs(gathered, self.nx)

    def sweep_foci(self, arr, foci_m, c, apod_args, origin_m, spacing_m, n, freq, rho, p0_pa):
        """every rank solves + accumulates its block of foci over the whole grid; returns |p| [F, nx, ny, nz] on
        every rank, in the caller's focus order."""
        self.plan_foci_sweep(arr, foci_m, c, apod_args, origin_m, spacing_m, n, freq, rho, p0_pa)
        self.step("allgather")
        return self.fetch_all()

    def aggregate(self):
        """Cross-rank aggregate of the volumes the last sweep left resident: (max_f |p|, mean_f I) over ALL
        ranks' GENUINE foci (padding excluded; plan/protocol.py:382-387) -- one RCCL all-reduce per volume
        instead of a gather."""
        ctx = self.engine.ctx
        if not self.comm:
            return ctx.field_aggregate(want_intensity=bool(ctx._flags & 2))
        ctx.field_allreduce_aggregate()
        return ctx.aggregate_fetch(want_intensity=bool(ctx._flags & 2))

    def sweep_slabs(self, arr, delays, apod, origin_m, spacing_m, n, freq, c, rho, p0_pa, medium=None):
        """every rank accumulates ALL foci over its x-slab; returns |p| [F, nx, ny, nz] on every rank."""
        self.plan_slab_sweep(arr, delays, apod, origin_m, spacing_m, n, freq, c, rho, p0_pa, medium=medium)
        self.step("allgather")
        return self.fetch_all()
