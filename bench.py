#!/usr/bin/env python3
"""Headline benchmark: pressure-field accumulate throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ONE launch of kernel 2 that
accumulates the complex pressure of this GPU's `--foci-per-gpu` foci (default 8 = one GPU's share of
BASELINE.json's 64-focus Wheel sweep, configs[2]) of a 256-element matrix array over a 256^3 grid,
with the element table and the steering table already resident in HBM.  `--foci-per-gpu 1` is the
single-focus accumulate.  For N > 1 (launched by torch.distributed.run, one rank per GPU) rank r takes
foci [8r, 8r+8) of the sweep (weak scaling; the compute needs no collective).  `--reassemble` selects
what crosses xGMI per step, on a side stream overlapped with the next step's compute:
  aggregate (default)  local max/mean over the rank's foci + RCCL reduce-scatter of ONE volume pair (rank r ends
                       up owning its 1/N of the global aggregate; olx_field_allreduce_aggregate would replicate it) --
                       the aggregated result of Protocol.calc_solution (plan/protocol.py:382-387), the only
                       cross-rank dependency the sharded path has;
  allgather            every per-focus |p| volume to every rank (north_star's reassembly; 67 MB per
                       focus per peer: xGMI-bound);
  none                 volumes stay sharded in each rank's HBM.
torch is used only for the rendezvous / barrier (gloo); the product path is ctypes -> HIP.

Prints ONE JSON line (rank 0).  `value` = V * N_el * F_total / time [Mvoxel-elements/s].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "openlifu-python_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# measured on MI355X by tools/ubench_valu.hip (profiles/ubench_valu_r01.txt), cycles per
# wave-instruction per SIMD at 8 waves/SIMD: plain f32 VALU 2.45, v_sin/v_cos/v_rsq 8.17
CYC_PLAIN, CYC_TRANS = 2.45, 8.17
N_SIMD, CLK_GHZ = 1024, 2.4


def synthetic_workload(grid_n: int, spacing_mm: float, el=(16, 16), pitch_mm=3.0, n_foci=1, seed=0):
    """SURVEY 8(d) inputs: flat matrix array (gen_matrix_array semantics), cubic grid centred in x,y,
    z from 5 mm, foci = Wheel(center, 5 mm) about (0,0,40) mm when n_foci > 1."""
    import openlifu_amd as ol
    arr = ol.Transducer.gen_matrix_array(nx=el[0], ny=el[1], pitch=pitch_mm, kerf=0.1 * pitch_mm, units="mm",
                                         sensitivity=1e5)
    half = (grid_n - 1) / 2 * spacing_mm
    setup = ol.SimSetup(spacing=spacing_mm, x_extent=(-half, half), y_extent=(-half, half),
                        z_extent=(5.0, 5.0 + (grid_n - 1) * spacing_mm))
    target = ol.Point(position=(0, 0, 40), units="mm")
    # BASELINE configs[2]: Wheel(center, 63 spokes, 5 mm) = 64 foci; rank r owns foci [r*F, r*F + F) (mod 64)
    sweep = ol.focal_patterns.Wheel(center=True, num_spokes=63, spoke_radius=5.0).get_targets(target)
    # shard order: centre, spoke 0, then mirror partners (i, 63 - i) side by side, so that a shard holds whole
    # mirror orbits -- their steering vectors coincide up to the array's symmetry and kernel 2c accumulates each once
    order = [0, 1] + [k for i in range(1, 32) for k in (1 + i, 1 + 63 - i)]
    foci = [sweep[order[(seed * n_foci + k) % len(order)]] for k in range(n_foci)]
    return arr, setup, foci


def cpu_baseline(arr, setup, foci, budget_s: float):
    """Times the fp64 NumPy oracle (oracle/field_oracle.py, kind "port": the reference has no NumPy
    field code to time) on a bounded centred sub-cube of the same workload, single thread."""
    from oracle import bf_oracle as bo, field_oracle as fo
    pos_m, _, area, _, _ = arr.element_table()
    d, a = bo.beamform(pos_m, np.zeros_like(pos_m), foci[0].get_position(units="m"), 1500.0)
    coords = [np.asarray(c.data) * 1e-3 for c in setup.get_coords().values()]

    def sub(n):
        return [c[(len(c) - n) // 2:(len(c) - n) // 2 + n] for c in coords]

    t = time.perf_counter()
    fo.field_on_grid(*sub(16), pos_m, area, d, a, 400e3, 1500.0, 1e5)
    rate = 16 ** 3 * len(pos_m) / (time.perf_counter() - t)
    n = int(min(len(coords[0]), max(16, round((budget_s * rate / len(pos_m)) ** (1 / 3)))))
    t = time.perf_counter()
    fo.field_on_grid(*sub(n), pos_m, area, d, a, 400e3, 1500.0, 1e5)
    dt = time.perf_counter() - t
    return {"value": n ** 3 * len(pos_m) / dt / 1e6, "unit": "Mvoxel-elements/s", "cores": 1, "kind": "port",
            "sample": f"fp64 NumPy oracle, centred {n}^3 sub-cube x {len(pos_m)} elements, 1 focus, {dt:.1f} s"}


def cpu_baseline_c(arr, setup, foci, budget_s: float):
    """Same definition in C + OpenMP on every host core (oracle/field_oracle.c)."""
    from oracle import bf_oracle as bo, c_oracle as co
    pos_m, _, area, _, _ = arr.element_table()
    d, a = bo.beamform(pos_m, np.zeros_like(pos_m), foci[0].get_position(units="m"), 1500.0)
    coords = [np.asarray(c.data) * 1e-3 for c in setup.get_coords().values()]
    n = min(len(coords[0]), 64)
    sub = [c[(len(c) - n) // 2:(len(c) - n) // 2 + n] for c in coords]
    t = time.perf_counter()
    co.field_on_grid(*sub, pos_m, area, d, a, 400e3, 1500.0, 1e5)
    rate = n ** 3 * len(pos_m) / (time.perf_counter() - t)
    n2 = int(min(len(coords[0]), max(n, round((budget_s * rate / len(pos_m)) ** (1 / 3)))))
    sub = [c[(len(c) - n2) // 2:(len(c) - n2) // 2 + n2] for c in coords]
    t = time.perf_counter()
    co.field_on_grid(*sub, pos_m, area, d, a, 400e3, 1500.0, 1e5)
    dt = time.perf_counter() - t
    return {"value": n2 ** 3 * len(pos_m) / dt / 1e6, "unit": "Mvoxel-elements/s", "cores": co.max_threads(),
            "kind": "port", "sample": f"fp64 C/OpenMP oracle, centred {n2}^3 sub-cube x {len(pos_m)} elements, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--clock-ramp-ms", type=float, default=60.0,
                    help="before the W warm-up steps, keep the GPU busy this long so that DVFS has left the idle clock "
                         "(the first ~20 launches after idle run 15-20 %% slower, tools/launch_series.py); 0 disables")
    ap.add_argument("--foci-per-gpu", type=int, default=8)
    ap.add_argument("--reassemble", choices=["aggregate", "allgather", "none"], default="aggregate")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--spacing-mm", type=float, default=0.25)
    ap.add_argument("--elements", type=str, default="16x16")
    ap.add_argument("--pitch-mm", type=float, default=3.0)
    ap.add_argument("--force-comm", action="store_true", help="exercise the RCCL path even with 1 rank")
    ap.add_argument("--medium", choices=["water", "skull"], default="water",
                    help="skull: BASELINE configs[4] synthetic skull-slab mask, heterogeneous layered-ray kernel")
    ap.add_argument("--device", type=int, default=None, help="HIP device for every rank (debug: oversubscribe one GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the cpu_baseline leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import openlifu_amd as ol  # loads libolx.so (system ROCm runtime) before any torch import
    from openlifu_amd import _native as nat
    from openlifu_amd.engine import grid_from_coords

    dist = None
    if world > 1:
        import torch.distributed as dist  # rendezvous + barrier only (gloo, CPU); RCCL is driven by libolx
        # gloo reports its connections on C stdout; stdout belongs to the one JSON line, so it points at stderr meanwhile
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()  # forces the pair connections (and their messages) now
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    el = tuple(int(v) for v in args.elements.split("x"))
    arr, setup, foci = synthetic_workload(args.grid, args.spacing_mm, el, args.pitch_mm, args.foci_per_gpu, seed=rank)
    F, N = len(foci), arr.numelements()
    eng = ol.get_engine(local_rank if args.device is None else args.device)
    ctx = eng.ctx
    eng.bind(arr)
    ctx.bf_solve(np.array([f.get_position(units="m") for f in foci]), 1500.0)  # kernel 1: steering stays resident
    gather = (world > 1 or args.force_comm) and args.reassemble != "none"
    gather_note = None
    if gather:
        ok = 1
        try:
            uid = [ctx.comm_unique_id() if rank == 0 else None]  # (libolx keeps RCCL's banner off stdout)
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], world, rank)
        except Exception as e:  # noqa: BLE001 - report, keep measuring the sharded compute
            ok, gather_note = 0, f"RCCL init failed: {e}"
        if dist is not None:  # every rank must take the same branch, or the collectives below would hang
            import torch
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0 and ok:
                gather_note = "RCCL init failed on another rank"
                ctx.comm_destroy()
            ok = int(flag[0])
        gather = bool(ok)
    origin, spacing, n = grid_from_coords(setup.get_coords())
    ctx.field_plan(origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, flags=nat.OUT_PMAG | nat.OUT_INTENSITY)
    V = int(np.prod(n))
    if args.medium == "skull":  # 8 mm <= z < 14 mm + 2 mm sin(2 pi x / 40 mm) cos(2 pi y / 40 mm)
        xs, ys, zs = (np.asarray(c.data, dtype=np.float32) * 1e-3 for c in setup.get_coords().values())
        zsurf = 14e-3 + 2e-3 * np.sin(2 * np.pi * xs / 40e-3)[:, None] * np.cos(2 * np.pi * ys / 40e-3)[None, :]
        skull = (zs[None, None, :] >= 8e-3) & (zs[None, None, :] < zsurf[:, :, None])
        ctx.field_set_medium(np.where(skull, 2800.0, 1500.0).astype(np.float32), np.where(skull, 6.0, 0.0).astype(np.float32),
                             np.where(skull, 1900.0, 1000.0).astype(np.float32))
        del skull

    def step():
        ctx.field_launch()
        if gather and args.reassemble == "aggregate":
            ctx.field_reduce_scatter_aggregate()
        elif gather:
            ctx.field_allgather()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    if args.clock_ramp_ms > 0:  # not steps: the same launches, discarded, until the shader clock has ramped up
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.clock_ramp_ms:
            ctx.field_launch()
            ctx.sync()
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = ctx.profile_end()
    compute_only = None
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        if gather:  # reported beside the headline (never instead of it): the same steps without the exchange
            k2 = min(args.steps, 200)
            barrier()
            t1 = time.perf_counter()
            for _ in range(k2):
                ctx.field_launch()
            barrier()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            compute_only = {"steps": k2, "ms_per_step": float(t[0]) / k2 * 1e3}

    if rank == 0:
        pairs_per_step = float(V) * N * F * world
        value = pairs_per_step * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (field_accum_k): algorithmic HBM bytes per launch =
        # 8 B per voxel per focus (|p| + intensity float32 outputs) + the 32 B/entry packed table
        alg_bytes = 8.0 * V * F + 32.0 * N * F
        k_ms = float(np.mean(kern_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            key = f"{args.elements}_{args.grid}_{F}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        # VALU-issue ceiling of the kernel variant in use: how many logical (voxel, element,
        # focus) pairs one evaluation of the geometry term G serves, and what that evaluation costs to issue.
        import re
        name = ctx.field_variant()
        m = re.search(r"mx(\d),my(\d),dx(\d),dy(\d),nf(\d+)", name)
        mm = re.search(r"field_mfma_k<mt\d+,nt(\d+),.*> (\d+) columns for (\d+) foci x (\d+) images in (\d+) tile", name)
        ml = re.search(r"field_(?:lattice|coset)_k<.*> (\d+) columns for (\d+) foci x (\d+) images in (\d+) tile.* (\d+) MFMA/launch", name)
        if ml:  # kernel 2d: the table arithmetic is amortised over 8 voxel rows x 64 elements; the matrix pipe is the ceiling.
            # On this chip MFMA and VALU issue add up (tools/ubench_clock.hip), so the MFMA-only time is a strict floor.
            n_mfma = int(ml.group(5))
            floor_s = n_mfma * 16.0 / (N_SIMD * CLK_GHZ * 1e9)
            ceil_pairs = float(V) * N * F / floor_s
            cyc_per_g, pairs_per_g = None, None
            what = ("1 fp16 hi*hi product per K-step + one K=128 e4m3 instruction (2 units) per two K-steps for both hi/lo corrections"
                    if "fp8corr" in name else "3 fp16 hi/lo products")
            model = (f"{n_mfma} matrix-pipe units of one v_mfma_f32_16x16x32_f16 per launch ({what}, padded row tiles included) at "
                     f"16 cycles each on {N_SIMD} SIMDs @{CLK_GHZ} GHz = {floor_s * 1e3:.3f} ms if nothing else issued")
        elif mm:  # kernel 2c: per G 5 plain + 3 transcendental + 4 half-rate (hi/lo split) VALU instructions, plus the
            # issue slots its share of the 3*NT MFMAs blocks (8 cycles each, 0.75*NT MFMAs per 64 terms)
            ntc, _, nfoci, nimg, ntile = (int(v) for v in mm.groups())
            cyc_per_g = 5 * CYC_PLAIN + 3 * CYC_TRANS + 4 * 4.2 + 6.0 * ntc
            pairs_per_g = nfoci * nimg / ntile
            model = (f"per G: 5 plain @{CYC_PLAIN} + 3 transcendental @{CYC_TRANS} + 4 half-rate @4.2 + {0.75 * ntc:.2f} MFMA "
                     f"issue slots @8 cycles, serving {pairs_per_g:.0f} pairs")
        elif m:
            mx, my, dx, dy, nf = (int(v) for v in m.groups())
            cyc_per_g = 4 * CYC_PLAIN + 3 * CYC_TRANS + dx * dy * nf * 2 * 4.2
            pairs_per_g, model = mx * my * nf, (f"per G: 4 plain @{CYC_PLAIN} + 3 transcendental @{CYC_TRANS} + "
                                               f"{dx * dy * nf} x 2 v_pk_fma @4.2 cycles, serving {mx * my * nf} pairs")
        else:
            cyc_per_g, pairs_per_g, model = 6 * CYC_PLAIN + 3 * CYC_TRANS, 1, f"per pair: 6 plain @{CYC_PLAIN} + 3 transcendental @{CYC_TRANS} cycles"
        if cyc_per_g is not None:
            ceil_pairs = N_SIMD * CLK_GHZ * 1e9 * 64 * pairs_per_g / cyc_per_g
        out = {
            "metric": "Mvoxel-elements/s pressure-field accumulate", "value": value, "unit": "Mvoxel-elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N}-element {args.elements} matrix array x {args.grid}^3 grid "
                                   f"({args.spacing_mm} mm), {F} foci per GPU of the 64-focus Wheel sweep "
                                   f"(BASELINE configs[2] shard), |p|+intensity out",
                       "elements": N, "grid": [int(v) for v in n], "foci_per_gpu": F, "frequency_hz": 400e3,
                       "medium": args.medium, "clock_ramp_ms": args.clock_ramp_ms,
                       "kernel": ctx.field_variant(),
                       "reassembly": (f"rccl-{args.reassemble}-overlapped" if gather else
                                      ("none" if (world == 1 or args.reassemble == "none") else "skipped")),
                       **({"reassembly_note": gather_note} if gather_note else {}),
                       **({"without_exchange": dict(compute_only, value=float(V) * N * F * world /
                                                    (compute_only["ms_per_step"] * 1e-3) / 1e6)} if compute_only else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms_avg": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the accumulate is issue bound (matrix pipe + VALU), not HBM bound "
                                 "(SURVEY 8(d), DESIGN.md 5); see issue_ceiling"},
            "issue_ceiling": {"achieved_Mpairs_s": float(V) * N * F / (k_ms * 1e-3) / 1e6, "peak_Mpairs_s": ceil_pairs / 1e6,
                             "frac": float(V) * N * F / (k_ms * 1e-3) / ceil_pairs,
                             "model": model + ("" if ml else "; per 64 lanes per SIMD, 1024 SIMDs @2.4 GHz (tools/ubench_valu.hip)")},
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(arr, setup, foci, args.cpu_seconds)
            out["cpu_baseline_c"] = cpu_baseline_c(arr, setup, foci, min(args.cpu_seconds, 10.0))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        if gather:
            ctx.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
