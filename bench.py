#!/usr/bin/env python3
"""Headline benchmark: pressure-field accumulate throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ONE launch of kernel 2 that accumulates the
complex pressure of this GPU's `--foci-per-gpu` foci (default 8 = one GPU's share of BASELINE.json's 64-focus Wheel
sweep, configs[2]) of a 256-element matrix array over a 256^3 grid, element table and steering table already resident
in HBM, plus -- for N > 1 -- the reassembly exchange.  The foci of a rank are whatever the PRODUCT's shard planner
(openlifu_amd.dist.plan_foci_orbits: whole mirror orbits per GPU) assigns it, driven through the product's
`ShardedField` (plan_foci_sweep / step); `--foci-per-gpu 1` is the single-focus accumulate (configs[1] / configs[3]).

For N > 1 (launched by torch.distributed.run, one rank per GPU; weak scaling: 8 foci per GPU, 64 at N = 8) the timed
step uses `--reassemble aggregate` by default -- the reduce-scatter of max |p| / mean intensity (plan/protocol.py:382-387), the
one cross-rank dependency of calc_solution's result, on a side stream, overlapped with the next step's compute; the same run
then reports north_star's `allgather` (every per-focus |p| volume to every rank over RCCL/xGMI: 3.8 GB inbound per rank and
step at N = 8, link-bound by construction) and the compute without any exchange beside it.  A line whose
exchange did not run as asked carries "degraded": true and the process exits non-zero.
torch is used only for the rendezvous / barrier (gloo); the product path is ctypes -> HIP.

Arithmetic: fp32 accumulate of fp16 hi/lo operands on the matrix cores.  `value` is measured in the LIBRARY DEFAULT mode
(`--corrections auto`): what Protocol.calc_solution runs unless told otherwise -- on this shard the two hi x lo correction products
go through e4m3 (<= 7.5e-6 of the volume maximum, include/olx.h; north_star's gate is 1e-5).  `--corrections fp16` times the opted-out mode (plan flag
OLX_FIELD_FP16_CORRECTION / SimSetup.options["fp8_correction"] = "0": three fp16 products, <= 2e-6) as the headline instead.  At
N = 1 the line carries the other mode beside it (`precision_safe` / `library_default`), a `parity` block measured in this run against the fp64 C oracle, and the other
shapes of the path (`legs`: single focus on / off axis, off-axis shard, 64-focus sweep), each planned, clock-ramped and timed
over its own >= 200 steps; `scans` = GB/s of the HBM-bound streaming kernels; `kernel1` with its CPU baselines.

Prints ONE JSON line (rank 0).  `value` = V * N_el * F_total / time [Mvoxel-elements/s].
"""
from __future__ import annotations

import argparse
import json
import os
import re
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
F0, C0, RHO0, SENS = 400e3, 1500.0, 1000.0, 1e5


def synthetic_workload(grid_n: int, spacing_mm: float, el=(16, 16), pitch_mm=3.0, offset_mm=(0.0, 0.0)):
    """SURVEY 8(d) inputs: flat matrix array (gen_matrix_array semantics), cubic grid centred in x,y, z from 5 mm,
    BASELINE configs[2]'s 64-focus sweep = Wheel(center, 63 spokes, 5 mm) about (0,0,40) mm (+ offset_mm)."""
    import openlifu_amd as ol
    arr = ol.Transducer.gen_matrix_array(nx=el[0], ny=el[1], pitch=pitch_mm, kerf=0.1 * pitch_mm, units="mm",
                                         sensitivity=SENS)
    half = (grid_n - 1) / 2 * spacing_mm
    setup = ol.SimSetup(spacing=spacing_mm, x_extent=(-half, half), y_extent=(-half, half),
                        z_extent=(5.0, 5.0 + (grid_n - 1) * spacing_mm))
    target = ol.Point(position=(offset_mm[0], offset_mm[1], 40), units="mm")
    pattern = ol.focal_patterns.Wheel(center=True, num_spokes=63, spoke_radius=5.0)
    return arr, setup, target, pattern


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pos_m, area, coords, focus_m, budget_s: float):
    """Times the fp64 NumPy oracle (oracle/field_oracle.py, kind "port": the reference has no NumPy
    field code to time) on a bounded centred sub-cube of the same workload, single thread."""
    from oracle import bf_oracle as bo, field_oracle as fo
    d, a = bo.beamform(pos_m, np.zeros_like(pos_m), focus_m, C0)

    def sub(n):
        return [c[(len(c) - n) // 2:(len(c) - n) // 2 + n] for c in coords]

    t = time.perf_counter()
    fo.field_on_grid(*sub(16), pos_m, area, d, a, F0, C0, SENS)
    rate = 16 ** 3 * len(pos_m) / (time.perf_counter() - t)
    n = int(min(len(coords[0]), max(16, round((budget_s * rate / len(pos_m)) ** (1 / 3)))))
    t = time.perf_counter()
    fo.field_on_grid(*sub(n), pos_m, area, d, a, F0, C0, SENS)
    dt = time.perf_counter() - t
    return {"value": n ** 3 * len(pos_m) / dt / 1e6, "unit": "Mvoxel-elements/s", "cores": 1, "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"fp64 NumPy oracle, centred {n}^3 sub-cube x {len(pos_m)} elements, 1 focus, {dt:.1f} s"}


def cpu_baseline_c(pos_m, area, coords, focus_m, budget_s: float):
    """Same definition in C + OpenMP on every host core (oracle/field_oracle.c)."""
    from oracle import bf_oracle as bo, c_oracle as co
    d, a = bo.beamform(pos_m, np.zeros_like(pos_m), focus_m, C0)
    n = min(len(coords[0]), 64)
    sub = [c[(len(c) - n) // 2:(len(c) - n) // 2 + n] for c in coords]
    t = time.perf_counter()
    co.field_on_grid(*sub, pos_m, area, d, a, F0, C0, SENS)
    rate = n ** 3 * len(pos_m) / (time.perf_counter() - t)
    n2 = int(min(len(coords[0]), max(n, round((budget_s * rate / len(pos_m)) ** (1 / 3)))))
    sub = [c[(len(c) - n2) // 2:(len(c) - n2) // 2 + n2] for c in coords]
    t = time.perf_counter()
    co.field_on_grid(*sub, pos_m, area, d, a, F0, C0, SENS)
    dt = time.perf_counter() - t
    return {"value": n2 ** 3 * len(pos_m) / dt / 1e6, "unit": "Mvoxel-elements/s", "cores": co.max_threads(),
            "kind": "port", "cpu_model": cpu_model(),
            "sample": f"fp64 C/OpenMP oracle, centred {n2}^3 sub-cube x {len(pos_m)} elements, {dt:.1f} s"}


def kernel1_cpu(arr, foci_m):
    """CPU legs of kernel 1 on this box: the vectorised fp64 restatement and the per-element Python loop that
    mirrors the reference's cost model (bf/delay_methods/direct.py:35), one thread, same F x N solve."""
    from oracle import bf_oracle as bo
    pos_m, _, _, _, _ = arr.element_table()
    ori = np.array([el.orientation for el in arr.elements], dtype=np.float64)
    pos_u = np.array([el.position for el in arr.elements], dtype=np.float64)
    units = arr.elements[0].units

    def best(fn, reps):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return min(ts) * 1e6
    vec = best(lambda: [bo.beamform(pos_m, ori, f, C0) for f in foci_m], 5)
    loop = best(lambda: [bo.beamform_per_element(pos_u, ori, f, C0, units=units) for f in foci_m], 2)
    return {"cpu_vectorised_us_per_solve": vec, "cpu_per_element_loop_us_per_solve": loop, "cpu_cores": 1, "cpu_model": cpu_model(),
            "cpu_what": "oracle/bf_oracle.py beamform (vectorised fp64 NumPy) and beamform_per_element (one Python call per element, "
                        "the cost model of bf/delay_methods/direct.py:35), Direct delays + Uniform apodization, best of 5 / 2"}


def sampled_parity(ctx, coords, pos_m, area, foci_m, n_samples=20000, check=(0, 1, 4), apod=("uniform", 1.0, 0.0), ori=None,
                   full_volume_test="tests/test_gpu_field.py::test_headline_shard_256cubed_full_volume_parity"):
    """Post-timing parity of the resident result against the fp64 C oracle: `n_samples` random voxels of up to three
    focus volumes, error normalised by each focus' own peak (the volume maximum for a focus inside the grid)."""
    from oracle import bf_oracle as bo, c_oracle as co
    xs, ys, zs = coords
    rng = np.random.default_rng(147)
    idx = np.stack([rng.integers(0, len(c), n_samples) for c in coords], axis=1)
    pts = np.stack([xs[idx[:, 0]], ys[idx[:, 1]], zs[idx[:, 2]]], axis=1)
    worst, checked = 0.0, []
    for f in check:
        if f >= len(foci_m):
            continue
        d, a = bo.beamform(pos_m, np.zeros_like(pos_m) if ori is None else ori, foci_m[f], C0, apod=apod)
        dmin = 0.5 * min(float(c[1] - c[0]) for c in coords)      # the definition's clamp (a grid through the element plane has voxels on elements)
        ref = np.abs(co.field_at_points(pts, pos_m, area, d, a, F0, C0, SENS, dmin=dmin))
        peak = max(np.abs(co.field_at_points([foci_m[f]], pos_m, area, d, a, F0, C0, SENS, dmin=dmin))[0], ref.max())
        got = ctx.field_fetch(f, want=("pmag",))["pmag"][idx[:, 0], idx[:, 1], idx[:, 2]]
        worst = max(worst, float(np.abs(got - ref).max() / peak))
        checked.append(int(f))
    return {"max_err_over_peak": worst, "gate": 1e-5, "sampled_voxels_per_focus": n_samples, "foci_checked": checked,
            "oracle": "oracle/field_oracle.c (fp64, C/OpenMP)",
            "full_volume_test": full_volume_test}


def issue_model(name, V, N, F):
    """Issue ceiling of the kernel variant in use: (peak pairs/s, model text)."""
    m = re.search(r"mx(\d),my(\d),dx(\d),dy(\d),nf(\d+)", name)
    mm = re.search(r"field_mfma_k<mt\d+,nt(\d+),.*> (\d+) columns for (\d+) foci x (\d+) images in (\d+) tile", name)
    ml = re.search(r"field_(?:lattice|coset|cosetp|toep)_k<.*> (\d+) columns for (\d+) foci x (\d+) images in (\d+) tile.* (\d+) MFMA/launch", name)
    if ml:  # lattice kernels: table arithmetic amortised; the matrix pipe is the ceiling (MFMA and VALU issue add up here)
        n_mfma = int(ml.group(5))
        floor_s = n_mfma * 16.0 / (N_SIMD * CLK_GHZ * 1e9)
        what = ("1 fp16 hi*hi product per K-step + one K=128 e4m3 instruction (2 units) per two K-steps for both hi/lo corrections"
                if "fp8corr" in name else "3 fp16 hi/lo products")
        return float(V) * N * F / floor_s, (f"{n_mfma} matrix-pipe units of one v_mfma_f32_16x16x32_f16 per launch ({what}, padded row tiles "
                                            f"included) at 16 cycles each on {N_SIMD} SIMDs @{CLK_GHZ} GHz = {floor_s * 1e3:.3f} ms if nothing else issued")
    if mm:
        ntc, _, nfoci, nimg, ntile = (int(v) for v in mm.groups())
        cyc, pairs = 5 * CYC_PLAIN + 3 * CYC_TRANS + 4 * 4.2 + 6.0 * ntc, nfoci * nimg / ntile
        model = (f"per G: 5 plain @{CYC_PLAIN} + 3 transcendental @{CYC_TRANS} + 4 half-rate @4.2 + {0.75 * ntc:.2f} MFMA "
                 f"issue slots @8 cycles, serving {pairs:.0f} pairs")
    elif m:
        mx, my, dx, dy, nf = (int(v) for v in m.groups())
        cyc, pairs = 4 * CYC_PLAIN + 3 * CYC_TRANS + dx * dy * nf * 2 * 4.2, mx * my * nf
        model = (f"per G: 4 plain @{CYC_PLAIN} + 3 transcendental @{CYC_TRANS} + {dx * dy * nf} x 2 v_pk_fma @4.2 cycles, "
                 f"serving {pairs} pairs")
    else:
        cyc, pairs, model = 6 * CYC_PLAIN + 3 * CYC_TRANS, 1, f"per pair: 6 plain @{CYC_PLAIN} + 3 transcendental @{CYC_TRANS} cycles"
    return N_SIMD * CLK_GHZ * 1e9 * 64 * pairs / cyc, model + "; per 64 lanes per SIMD, 1024 SIMDs @2.4 GHz (tools/ubench_valu.hip)"


def mfma_useful(name: str) -> dict:
    """{"mfma_issued", "mfma_dense", "mfma_useful"} for the lattice kernels (their planner states both counts in units of one
    v_mfma_f32_16x16x32_f16): useful = what the dense contraction needs / what the launch issues -- row, column and K-slot padding and the
    Toeplitz band of kernel 2f all show up here."""
    m = re.search(r"(\d+) MFMA/launch \((\d+) dense\)", name)
    if not m:
        return {}
    issued, dense = int(m.group(1)), int(m.group(2))
    return {"mfma_issued": issued, "mfma_dense": dense, "mfma_useful": dense / issued if issued else None}


def _library_stamp(nat):
    """The commit libolx.so was built from (openlifu-python_amd/build.py writes <library>.stamp; the GPU box has no .git)."""
    try:
        with open(nat.LIB_PATH + ".stamp") as f:
            return f.read().strip()
    except OSError:
        return "unknown"


def static_traffic(kernel_name: str, grid_n: int):
    """HBM bytes per launch from the committed PMC summaries (profiles/traffic.json), matched on the kernel variant
    string: a STATIC figure from tools/profile_round.sh's counter passes, not measured in this run; None when the
    variant in use has no committed counter pass."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    for ent in json.load(open(tpath)).get("entries", []):
        if ent.get("grid") == grid_n and kernel_name.startswith(ent.get("kernel_prefix", "\0")):
            return ent["hbm_bytes_per_launch"], f"static: {ent['source']} (rocprofv3 --pmc passes of tools/profile_round.sh on this kernel variant, not this run)"
    return None, None


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N ...` without a launcher around it: start N fresh rank processes of this same command line
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would, one device each), relay rank 0's
    JSON line, return the worst exit code.  The parent never imports the package and never touches HIP; children are started with
    subprocess (never exec'd over a process that holds a GPU) and, when one of them fails, the others get a grace period to reach
    their own error handling before they are terminated by PID."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: RCCL and the p2p transport both need it on this driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    import threading
    box = []
    reader = threading.Thread(target=lambda: box.append(procs[0].stdout.read()), daemon=True)   # rank 0 prints the one JSON line
    reader.start()
    grace_until = None                         # set when the first rank fails: the others get 60 s to fail on their own
    while any(p.poll() is None for p in procs):
        if grace_until is None and any(p.poll() not in (None, 0) for p in procs):
            grace_until = time.time() + 60.0
        if grace_until is not None and time.time() > grace_until:
            for p in procs:
                if p.poll() is None:
                    p.terminate()              # (by PID: these are our own children)
            t_kill = time.time() + 10.0
            while any(p.poll() is None for p in procs) and time.time() < t_kill:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    rcs = [p.wait() for p in procs]
    sys.stdout.write((box[0] if box else b"").decode(errors="replace"))
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        print(f"bench launcher: rank exit codes {rcs}", file=sys.stderr)
        return max(min(abs(rc), 255) for rc in bad)
    return 0


def pmc_pass(kernel_name: str, argv_workload, counters):
    """One child run of this same bench (same workload, 20 steps) under `rocprofv3 --kernel-trace --pmc <counters>`: ({counter: mean over the
    dominant kernel's dispatches}, mean dispatch duration [us] under the counters).  Raises on any failure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        raise RuntimeError("rocprofv3 not found")
    fn = kernel_name.split("<")[0].split(" ")[0]           # e.g. field_cosetp_k
    tmp = tempfile.mkdtemp(prefix="olx_pmc_", dir="/tmp")
    try:
        cmd = [prof, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__),
               "--no-extras", "--cpu-seconds", "0", "--steps", "20", "--warmup", "3"] + list(argv_workload)
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=240)
        if r.returncode != 0:
            raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode})")
        acc, dur, seen = {c: [] for c in counters}, [], set()
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if fn in row.get("Kernel_Name", "") and row.get("Counter_Name") in acc:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Dispatch_Id"] not in seen:
                        seen.add(row["Dispatch_Id"])
                        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        if any(len(v) < 5 for v in acc.values()):
            raise RuntimeError(f"no {fn} dispatches in the {' '.join(counters)} pass")
        return {c: sum(v) / len(v) for c, v in acc.items()}, sum(dur) / len(dur)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_traffic(kernel_name: str, argv_workload):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two child runs of this same bench (same workload, 20 steps) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` -- separate passes, as MI355X_MICROARCH.md prescribes -- and
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 averaged over the kernel's dispatches (gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes).
    Returns (bytes, source) or (None, reason); any failure (no rocprofv3, a pass that times out) leaves the static figure in place."""
    fn = kernel_name.split("<")[0].split(" ")[0]
    try:
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[ctr] = pmc_pass(kernel_name, argv_workload, [ctr])[0][ctr]
        nbytes = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        return nbytes, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate child passes of this bench, 20 steps each), "
                        f"(2 x {vals['FETCH_SIZE']:.0f} + {vals['WRITE_SIZE']:.0f}) KiB per {fn} launch; FETCH_SIZE doubled per the gfx950 correction")
    except Exception as e:  # noqa: BLE001 - the static figure stays
        return None, f"live counter passes failed: {e}"


N_CU = 256
STORE_DRAIN_GBS = 6300.0     # sustained write / stream-copy rate of this part (tools/ubench_store.hip: 5.1 - 6.6 TB/s; DESIGN.md 5.3: 6.3 TB/s)
STORE_RUN64_GBS = 4300.0     # ... and what it absorbs in scattered runs of 64 contiguous bytes, the lattice kernels' store pattern (one block = 16 planes of a
                             # voxel column; profiles/ubench_store_r05.txt: 64 B 4.3, 128 B 5.4, 1 KiB 7.7 TB/s)


def achievable(kernel_name: str, argv_workload, alg_bytes: float, k_ms: float):
    """What THIS FORMULATION permits, from counters measured in this run (one more child pass: SQ_BUSY_CU_CYCLES, SQ_INSTS_VALU, SQ_INSTS_MFMA,
    SQ_INSTS_VALU_TRANS_F32): a SIMD has ONE vector issue port that its matrix and its other vector instructions share, so
    the launch cannot be shorter than (16 cycles per matrix unit + 2.45 per other vector instruction, 8.17 per transcendental -- the issue costs
    tools/ubench_valu.hip measures) / 1024 SIMDs at the clock the chip holds under this load (power cap), nor than the HBM write drain of its
    result in the kernel's store pattern."""
    try:
        v, dur_us = pmc_pass(kernel_name, argv_workload, ["SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32"])
    except Exception as e:  # noqa: BLE001
        return {"skipped": str(e)}
    clk = v["SQ_BUSY_CU_CYCLES"] / N_CU / (dur_us * 1e-6) / 1e9
    n_mfma, n_other, n_trans = v["SQ_INSTS_MFMA"], v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"], v["SQ_INSTS_VALU_TRANS_F32"]
    # matrix-pipe time in units of one v_mfma_f32_16x16x32_f16 (16 cycles): the e4m3 instruction (K = 128) takes two units, so where the
    # planner states the launch's units (lattice kernels) they are used instead of the instruction count
    mu = re.search(r"(\d+) MFMA/launch", kernel_name)
    units = float(mu.group(1)) if mu else n_mfma
    # (issue costs measured by tools/ubench_valu.hip at four waves per SIMD: a plain fp32 instruction 2.45 cycles, a transcendental 8.17; conversions
    # and mixed-precision instructions cost up to 4.2 -- the cheapest figure keeps this a floor)
    matrix_cyc, other_cyc = 16.0 * units, CYC_PLAIN * (n_other - n_trans) + CYC_TRANS * n_trans
    matrix_ms = matrix_cyc / (N_SIMD * clk * 1e9) * 1e3
    port_ms = (matrix_cyc + other_cyc) / (N_SIMD * clk * 1e9) * 1e3
    runs64 = kernel_name.startswith(("field_cosetp_k", "field_coset_k", "field_toep_k"))
    store_rate = STORE_RUN64_GBS if runs64 else STORE_DRAIN_GBS
    store_ms = alg_bytes / (store_rate * 1e9) * 1e3
    floor_ms = max(port_ms, store_ms)
    return {"what": "floors of this formulation at the clock measured under this load: vector issue port of the SIMDs (matrix + other vector "
                    "instructions share it) and the HBM write drain of the result in the kernel's store pattern; `floor_ms` = the larger one "
                    "(perfect overlap of the two).  Lower bounds, not a model of the launch: round 5's A/B runs (profiles/r05_store_path.txt) "
                    "show the headline launch bound by neither alone -- 22 % fewer vector instructions left it as long as before, cache-resident "
                    "stores make it 0.305 ms; a block is a chain of barrier-separated phases on different units with two blocks per CU",
            "store_pattern": "runs of 64 contiguous bytes (16 planes of a voxel column per block)" if runs64 else "streaming",
            "clock_ghz": clk, "clock_source": "SQ_BUSY_CU_CYCLES / 256 CUs / dispatch duration, child pass of this run",
            "matrix_instructions": n_mfma, "matrix_units_of_16_cycles": units, "other_vector_instructions": n_other, "transcendentals": n_trans,
            "matrix_only_ms": matrix_ms, "issue_port_ms": port_ms, "store_drain_ms": store_ms, "store_drain_rate_GBps": store_rate,
            "floor_ms": floor_ms, "roofline_frac_at_floor": alg_bytes / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "measured_ms": k_ms, "measured_over_floor": k_ms / floor_ms,
            "busy_cycles_per_cu": v["SQ_BUSY_CU_CYCLES"] / N_CU,
            "ms_at_2p4ghz": v["SQ_BUSY_CU_CYCLES"] / N_CU / (CLK_GHZ * 1e9) * 1e3,
            "clock_note": "the chip holds `clock_ghz` under this load (power cap; 2.4 GHz nominal): the launch is `busy_cycles_per_cu` cycles long, `ms_at_2p4ghz` "
                          "is what those cycles would take at the nominal clock -- with cache-resident stores the same kernel runs 7 % fewer cycles at an 8 % higher clock "
                          "(profiles/r05_store_path.txt, 15)",
            "matrix_only_ms_at_2p4ghz": matrix_cyc / (N_SIMD * CLK_GHZ * 1e9) * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--clock-ramp-ms", type=float, default=60.0,
                    help="before the W warm-up steps, keep the GPU busy this long so that DVFS has left the idle clock "
                         "(the first ~20 launches after idle run 15-20 %% slower, tools/launch_series.py); 0 disables")
    ap.add_argument("--foci-per-gpu", type=int, default=8)
    ap.add_argument("--reassemble", choices=["allgather", "aggregate", "none"], default=None,
                    help="what the timed step exchanges.  Default for N > 1: aggregate -- the reduce-scatter of max |p| / mean intensity, the one cross-rank "
                         "dependency of calc_solution's result (plan/protocol.py:382-387) and the exchange that can scale on point-to-point xGMI "
                         "(DESIGN.md 6); north_star's all-gather of every per-focus volume and the compute without exchange are timed beside it.  N = 1: none")
    ap.add_argument("--corrections", choices=["auto", "fp8", "fp16"], default="auto",
                    help="hi x lo correction products of the fp16 operand split: auto (= fp8) = the library default, e4m3 products where their "
                         "bound (<= 7.5e-6 of the volume maximum, include/olx.h) is a bound on the planned volume; fp16 = opted out (plan flag "
                         "OLX_FIELD_FP16_CORRECTION, <= 2e-6); the other one is timed beside it at N = 1")
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--spacing-mm", type=float, default=0.25)
    ap.add_argument("--elements", type=str, default="16x16")
    ap.add_argument("--pitch-mm", type=float, default=3.0)
    ap.add_argument("--offset-mm", type=str, default="0,0",
                    help="lateral offset of the sweep's target: a non-zero value breaks the mirror symmetry the headline shard enjoys")
    ap.add_argument("--force-comm", action="store_true", help="exercise the RCCL path even with 1 rank")
    ap.add_argument("--gather", choices=["rccl", "p2p"], default="rccl",
                    help="transport of the reassembly (include/olx.h OLX_GATHER): rccl = ncclAllGather (north_star), p2p = direct pulls over "
                         "HIP IPC, one stream per peer; the other one is timed beside it (and takes over if this one cannot be initialised)")
    ap.add_argument("--medium", choices=["water", "skull"], default="water",
                    help="skull: BASELINE configs[4] synthetic skull-slab mask, heterogeneous layered-ray kernel, x-slabs per GPU")
    ap.add_argument("--hetero-planes-per-layer", type=int, default=1,
                    help="--medium skull: quadrature of the ray integrals (1 = one sample per grid plane, the default model; G > 1 = "
                         "opt-in layered screens, olx_field_medium_layering)")
    ap.add_argument("--hetero-model", choices=["auto", "marched", "sampled"], default="auto",
                    help="--medium skull: ray-integral model (olx_field_medium_model): marched = running ray sums, one look-up per ray "
                         "(kernel 2m, what auto picks for this phantom); sampled = one sample per non-trivial plane (kernel 2h)")
    ap.add_argument("--device", type=int, default=None, help="HIP device for every rank (debug: oversubscribe one GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the post-timing legs (other correction mode, parity, calc_solution)")
    ap.add_argument("--static-traffic", action="store_true", help="report roofline.traffic from profiles/traffic.json instead of measuring it in child rocprofv3 passes")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process becomes the launcher (it has not imported the package or touched HIP)
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        args.gpus = world
    reassemble = args.reassemble or ("aggregate" if world > 1 else "none")

    import openlifu_amd as ol  # loads libolx.so (system ROCm runtime) before any torch import
    from openlifu_amd import _native as nat, dist as od
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
    off = tuple(float(v) for v in args.offset_mm.split(","))
    arr, setup, target, pattern = synthetic_workload(args.grid, args.spacing_mm, el, args.pitch_mm, off)
    sweep = pattern.get_targets(target)
    sweep_m = np.array([f.get_position(units="m") for f in sweep])
    origin, spacing, n = grid_from_coords(setup.get_coords())
    coords_m = [np.asarray(c.data) * 1e-3 for c in setup.get_coords().values()]
    centre = tuple(origin[a] + 0.5 * (n[a] - 1) * spacing[a] for a in (0, 1))
    # weak scaling: the 64-focus sweep cut into orbit-aware shards of `foci_per_gpu`; an N-GPU run takes the first N of them
    fpg = max(1, min(args.foci_per_gpu, len(sweep_m)))
    shards_all = od.plan_foci_orbits(sweep_m, -(-len(sweep_m) // fpg), centre_xy=centre)
    run_idx = np.concatenate([shards_all[r % len(shards_all)] for r in range(world)])
    run_foci = sweep_m[run_idx]
    N = arr.numelements()
    V = int(np.prod(n))
    eng = ol.get_engine(local_rank if args.device is None else args.device)
    ctx = eng.ctx
    sf = od.ShardedField(eng, world, rank)
    gather = (world > 1 or args.force_comm) and reassemble != "none"
    gather_note = None

    def exchange(uid):
        box = [uid]
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        return box[0]

    def allgather_bytes(blob):
        if dist is None:
            return [blob]
        box = [None] * world
        dist.all_gather_object(box, blob)
        return box

    def init_transport(kind):
        """Initialise the reassembly transport `kind` ("rccl" | "p2p") on every rank; (ok on ALL ranks, note)."""
        ok, note = 1, None
        os.environ["OLX_GATHER"] = kind        # read by rank 0 when it makes the id; the id tells the others
        if stuck:       # an earlier init never came back: this context's communicator state is not ours to touch again
            ok, note = 0, f"{kind} not tried: {stuck[0]}"
        else:
            # The first multi-device ncclCommInitRank of this code happens on whatever node runs `--gpus N`: it runs on a watchdog thread, so
            # that an init that never returns costs the exchange, not the whole line (the timed step then runs without exchange and says so).
            import threading
            box = []

            def work():
                try:
                    sf.init_comm(exchange, allgather_bytes)  # (libolx keeps RCCL's banner off stdout)
                    box.append(None)
                except Exception as e:  # noqa: BLE001 - report, keep measuring the sharded compute
                    box.append(e)
            th = threading.Thread(target=work, daemon=True)
            th.start()
            th.join(float(os.environ.get("OLX_BENCH_COMM_TIMEOUT_S", "300")))
            if th.is_alive():
                stuck.append(f"{kind} init did not return within {os.environ.get('OLX_BENCH_COMM_TIMEOUT_S', '300')} s")
                ok, note = 0, stuck[0]
            elif box and box[0] is not None:
                ok, note = 0, f"{kind} init failed: {box[0]}"
        if dist is not None:  # every rank must take the same branch, or the collectives below would hang
            import torch
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0:
                if ok:
                    note = f"{kind} init failed on another rank"
                if not stuck:
                    sf.close()
            ok = int(flag[0])
        elif not ok and not stuck:
            sf.close()
        return bool(ok), note

    transport = None
    stuck = []       # set when a communicator init never returned (see init_transport)
    if gather:
        # north_star's reassembly is RCCL; the direct peer-to-peer all-gather (HIP IPC, all 7 xGMI links at once) is timed beside
        # it, and takes over as the primary transport when RCCL cannot be brought up
        order = [args.gather] + [k for k in ("rccl", "p2p") if k != args.gather]
        notes = []
        for kind in order:
            ok, note = init_transport(kind)
            if note:
                notes.append(note)
            if ok:
                transport = kind
                break
        gather = transport is not None
        gather_note = "; ".join(notes) if notes else None
    out_flags = nat.OUT_PMAG | nat.OUT_INTENSITY
    skull = None
    if args.medium == "skull":  # 8 mm <= z < 14 mm + 2 mm sin(2 pi x / 40 mm) cos(2 pi y / 40 mm)
        from openlifu_amd.seg.seg_methods import skull_slab_volumes
        skull = skull_slab_volumes(*coords_m)
        skull["planes_per_layer"] = args.hetero_planes_per_layer
        skull["model"] = args.hetero_model

    def plan(fp8: bool):
        if skull is not None:   # configs[4]: all foci of the run on every rank's x-slab, label volume replicated
            dl, ap = eng.beamform(arr, run_foci[:fpg], C0)            # kernel 1
            sf.plan_slab_sweep(arr, dl, ap, origin, spacing, n, F0, C0, RHO0, SENS, flags=out_flags, medium=skull)
            return fpg
        mine = sf.plan_foci_sweep(arr, run_foci, C0, (nat.APOD_UNIFORM, 1.0, 0.0), origin, spacing, n, F0, RHO0, SENS,
                                  flags=out_flags, fp8_correction=(None if fp8 else False))
        return len(mine)

    def agree(ok: bool) -> bool:
        """True only when EVERY rank says ok (gloo all-reduce): a leg that failed on one rank is abandoned by all of them together."""
        if dist is None:
            return ok
        import torch
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag[0]))

    def timed(mode: str, steps: int, warmup: int):
        """(wall seconds for `steps` steps: max over ranks, per-launch kernel ms of this rank), or None when the leg failed on some
        rank.  Device-side failures (an exchange that times out, a transport error) are caught per rank; the rendezvous calls below
        are reached by every rank whatever happened, so one rank's failure cannot leave the others waiting in a barrier."""
        err = None
        try:
            for _ in range(warmup):
                sf.step(mode)
            ctx.sync()
        except Exception as e:  # noqa: BLE001
            err = e
        if dist is not None:
            dist.barrier()
        kern, t0 = np.zeros(0, dtype=np.float32), time.perf_counter()
        try:
            if err is None:
                ctx.profile_begin(steps)
                t0 = time.perf_counter()
                for _ in range(steps):
                    sf.step(mode)
                ctx.sync()
        except Exception as e:  # noqa: BLE001
            err = e
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        try:
            kern = ctx.profile_end() if err is None else kern
        except Exception as e:  # noqa: BLE001
            err = e
        if not agree(err is None):
            mine = str(err) if err is not None else ""
            if dist is not None:        # every rank's own message (the first one to give up names the cause, the others only see the abort)
                box = [None] * world
                dist.all_gather_object(box, mine)
                mine = "; ".join(f"rank {i}: {e}" for i, e in enumerate(box) if e)
            timed.last_error = mine or "failed on another rank"
            return None
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return elapsed, kern
    timed.last_error = ""

    def ramp():
        """Before EVERY timed leg: the planned launch, discarded, until the shader clock has left the idle state (the GPU idles
        for seconds during planning, parity checks and CPU legs; the first ~20 launches after idle run 15-20 % slower)."""
        if args.clock_ramp_ms > 0:
            t_ramp = time.perf_counter()
            try:
                while (time.perf_counter() - t_ramp) * 1e3 < args.clock_ramp_ms:
                    ctx.field_launch()
                    ctx.sync()
            except Exception:  # noqa: BLE001 - a broken exchange shows up (and is handled) in the timed leg that follows
                pass

    F = plan(args.corrections != "fp16")
    ramp()
    mode = reassemble if gather else "none"
    res = timed(mode, args.steps, args.warmup)
    if res is None and mode != "none":      # the exchange failed: the line still reports the sharded compute, and says so
        gather_note = ((gather_note + "; ") if gather_note else "") + f"{transport} {mode} failed ({timed.last_error}); timed without exchange"
        gather, mode = False, "none"
        try:
            if not stuck:
                sf.close()
        except Exception:  # noqa: BLE001
            pass
        plan(args.corrections != "fp16")
        ramp()
        res = timed("none", args.steps, args.warmup)
    if res is None:
        sys.exit(f"bench: the accumulate itself failed: {timed.last_error}")
    elapsed, kern_ms = res
    try:
        ranks_seen = ctx.comm_ranks_seen() if gather else 0     # what RCCL / the p2p control block counted (0: no exchange ran)
    except Exception:  # noqa: BLE001
        ranks_seen = -1
    kernel_name = ctx.field_variant()
    k_ms = float(np.mean(kern_ms))
    F_total = F * world if skull is None else F   # slab mode: the same F foci on every rank's slab
    V_step = float(V)                              # voxels of the whole grid covered per step (slabs tile it)
    pairs_per_step = V_step * N * F_total
    value = pairs_per_step * args.steps / elapsed / 1e6
    beside = {}
    if dist is not None and gather:  # reported beside the headline (never instead of it): other exchange, no exchange
        k2 = min(args.steps, 200)
        other = "aggregate" if mode == "allgather" else "allgather"
        def beside_leg(key, m):
            ramp()
            r = timed(m, k2, 10)
            beside[key] = ({"steps": k2, "ms_per_step": r[0] / k2 * 1e3, "value": pairs_per_step * k2 / r[0] / 1e6} if r is not None
                           else {"skipped": timed.last_error})
            return r is not None
        if skull is None:
            beside_leg(f"with_{other}", other)
        if skull is None:      # the all-gather over the other transport
            other_t = "p2p" if transport == "rccl" else "rccl"
            os.environ.setdefault("OLX_P2P_TIMEOUT_S", "20")
            sf.close()
            ok_t, note_t = init_transport(other_t)
            if ok_t:
                plan(args.corrections != "fp16")
                if not beside_leg(f"with_{other_t}_allgather", "allgather"):
                    try:
                        sf.close()          # a failed transport is not used again
                    except Exception:  # noqa: BLE001
                        pass
                    plan(args.corrections != "fp16")
            else:
                beside[f"with_{other_t}_allgather"] = {"skipped": note_t}
        beside_leg("without_exchange", "none")

    if rank == 0:
        # roofline of the dominant kernel: algorithmic HBM bytes per launch = 8 B per voxel per focus (|p| + intensity
        # float32 outputs) + the 32 B/entry packed steering table; a slab launch covers V / world voxels
        vox_launch = V if skull is None else V // world
        alg_bytes = 8.0 * vox_launch * F + 32.0 * N * F
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = static_traffic(kernel_name, args.grid)
        ceil_pairs, model = issue_model(kernel_name, vox_launch, N, F)
        fp8_on = "fp8corr" in kernel_name
        lattice = "field_coset" in kernel_name or "field_lattice_k" in kernel_name or "field_toep" in kernel_name or "field_mfma_k" in kernel_name
        dtype = ("f32-acc/f16x2+e4m3-corr" if fp8_on else ("f32-acc/f16x3" if lattice else "f32"))
        out = {
            "metric": "Mvoxel-elements/s pressure-field accumulate", "value": value, "unit": "Mvoxel-elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if skull is None else "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "library_built_from_commit": _library_stamp(nat),
            # which exchange `value` includes: the scaling claim of this line (the other modes are reported beside it in `config`)
            "scaling_claim": mode if gather else "none",
            "n_ranks_rendezvous": world, "n_ranks_seen": ranks_seen if gather else world,
            "n_ranks_seen_by": (f"{transport} communicator" if gather else "process launcher (no exchange in the timed step)"),
            # a line whose exchange did not run as asked (a transport that never came up, fewer ranks counted than launched) says so and the
            # process exits non-zero after printing it
            "degraded": bool(stuck) or (world > 1 and reassemble != "none" and (not gather or ranks_seen != world)),
            "config": {"workload": f"{N}-element {args.elements} matrix array x {args.grid}^3 grid "
                                   f"({args.spacing_mm} mm), {F} foci per GPU of the 64-focus Wheel sweep "
                                   f"(BASELINE configs[2] shard, planned by openlifu_amd.dist.plan_foci_orbits), |p|+intensity out"
                                   if skull is None else
                                   f"{N}-element {args.elements} matrix array x {args.grid}^3 grid ({args.spacing_mm} mm), skull-slab "
                                   f"medium (BASELINE configs[4]), {F} foci, x-slabs of {n[0] // world} planes per GPU",
                       "elements": N, "grid": [int(v) for v in n], "foci_per_gpu": F, "frequency_hz": F0,
                       "focus_indices_rank0": [int(v) for v in run_idx[:F]], "target_offset_mm": list(off),
                       "medium": args.medium, "clock_ramp_ms": args.clock_ramp_ms,
                       "corrections": "e4m3 (library default where the foci lie in the planned volume and N_eff >= 256)" if fp8_on else ("fp16 (opted out: OLX_FIELD_FP16_CORRECTION)" if args.corrections == "fp16" else "fp16 (library default for this shape)"),
                       "kernel": kernel_name,
                       "reassembly": (f"{transport}-{mode}-overlapped" if gather else
                                      ("none" if (world == 1 or reassemble == "none") else "skipped")),
                       **({"rccl_library": ctx.rccl_path()} if gather and transport == "rccl" else {}),
                       **({"reassembly_note": gather_note} if gather_note else {}), **beside},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms_avg": k_ms, "kernel_launches_timed": int(len(kern_ms)),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the accumulate is bound on the compute side (a block's chain of vector-ALU, matrix pipe + LDS and store phases "
                                 "at the clock the chip holds under its power cap), not by HBM bandwidth: SURVEY 8(d), DESIGN.md 5.5; see "
                                 "`achievable` and `issue_ceiling`"},
            **mfma_useful(kernel_name),
            "issue_ceiling": {"achieved_Mpairs_s": float(vox_launch) * N * F / (k_ms * 1e-3) / 1e6, "peak_Mpairs_s": ceil_pairs / 1e6,
                              "frac": float(vox_launch) * N * F / (k_ms * 1e-3) / ceil_pairs, "model": model},
        }
        if world == 1 and not args.no_extras and not args.static_traffic:
            wl = ["--grid", str(args.grid), "--spacing-mm", str(args.spacing_mm), "--elements", args.elements, "--pitch-mm", str(args.pitch_mm),
                  "--offset-mm", args.offset_mm, "--foci-per-gpu", str(args.foci_per_gpu), "--corrections", args.corrections, "--medium", args.medium]
            nbytes, src = live_traffic(kernel_name, wl)
            if nbytes is not None:
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = nbytes, src
            else:
                out["roofline"]["traffic_source"] = (traffic_src or "none") + f" [live measurement unavailable: {src}]"
            out["achievable"] = achievable(kernel_name, wl, alg_bytes, k_ms)
        if world == 1 and not args.no_extras and skull is None:
            pos_m, _, area, _, _ = arr.element_table()
            out["parity"] = {("e4m3" if fp8_on else "fp16"): sampled_parity(ctx, coords_m, pos_m, area, run_foci[:F])}
            k2 = max(200, min(args.steps, 500))     # secondary legs: their own floor, whatever --steps says

            def leg(foci_m, fp8, what):
                """plan -> clock ramp -> 20 warm-up + k2 timed steps of one more shape of the path."""
                foci_m = np.atleast_2d(foci_m)
                sf.plan_foci_sweep(arr, foci_m, C0, (nat.APOD_UNIFORM, 1.0, 0.0), origin, spacing, n, F0, RHO0, SENS, flags=out_flags,
                                   fp8_correction=(None if fp8 else False))
                ramp()
                e, km = timed("none", k2, 20)       # (single rank: a failure here is a bug and raises below)
                name = ctx.field_variant()
                nf = foci_m.shape[0]
                bytes_l = 8.0 * V * nf + 32.0 * N * nf
                m = re.search(r"(\d+) columns for (\d+) foci x (\d+) images", name)
                return {**mfma_useful(name), "what": what, "kernel": name, "dtype": "f32-acc/f16x2+e4m3-corr" if "fp8corr" in name else "f32-acc/f16x3",
                        "foci": nf, "columns_computed": int(m.group(1)) if m else None,
                        "kernel_ms_avg": float(np.mean(km)), "kernel_launches_timed": int(len(km)), "ms_per_step": e / k2 * 1e3, "steps": k2,
                        "value": float(V) * N * nf * k2 / e / 1e6, "roofline_frac": bytes_l / (float(np.mean(km)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": bytes_l, "traffic": static_traffic(name, args.grid)[0]}

            # the other correction mode on the same workload, same box
            other_fp8 = args.corrections == "fp16"
            o = leg(run_foci[:F], other_fp8, "the headline shard in the other correction mode")
            out["library_default" if other_fp8 else "precision_safe"] = o
            out["parity"]["e4m3" if "fp8corr" in o["kernel"] else "fp16"] = sampled_parity(ctx, coords_m, pos_m, area, run_foci[:F])
            legs = {}
            if off == (0.0, 0.0):
                on_axis = sweep_m[0]                                          # the Wheel's centre = the target itself
                t2 = ol.Point(position=(1.3, 0.7, 40), units="mm")           # off the array axis: no (focus, image) pair shares a column
                sw2 = np.array([f.get_position(units="m") for f in pattern.get_targets(t2)])
                legs["single_focus_on_axis"] = leg(on_axis, True, "one focus on the array axis (SinglePoint, the reference's default pattern)")
                legs["single_focus_off_axis"] = leg(sw2[0], True, "one focus 1.3 / 0.7 mm off the array axis (an arbitrary target: 4 mirror images = 4 columns)")
                legs["asymmetric"] = leg(sw2[od.plan_foci_orbits(sw2, -(-len(sw2) // fpg), centre_xy=centre)[0]], True,
                                         "same shard size, sweep target offset by (1.3, 0.7) mm from the array axis: the mirror folds of the grid "
                                         "still apply, but no two (focus, image) pairs share a steering column")
                legs["sweep64"] = leg(sweep_m, True, "the whole 64-focus Wheel sweep of configs[2] on one GPU")
            legs.update(config_legs(ol, nat, od, eng, sf, timed, ramp, out_flags))
            out["legs"] = legs
            # HBM-bound streaming scans over the resident result of the headline shard
            plan(args.corrections != "fp16")
            ctx.field_launch(); ctx.sync()
            scans = {}
            for kname in ("aggregate", "scale", "analysis_peaks", "masked_peak", "weighted_sum", "offset_grid", "fused_post"):
                ctx.scan_time(kname, 5)                                       # warm
                ms, nbytes = ctx.scan_time(kname, 30)
                t_ms = float(np.mean(ms))
                scans[kname] = {"ms": t_ms, "algorithmic_bytes": nbytes, "GBps": nbytes / (t_ms * 1e-3) / 1e9,
                                "roofline_frac": nbytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["scans"] = {"what": f"streaming kernels over the {F} resident focus volumes of the headline shard (offset_grid: one fp64 grid), "
                                    "30 launches each, HIP events (olx_scan_time)", **scans}
            # kernel 1: microseconds per F x N solve, HIP events around 50 repeats, with its CPU restatements beside it
            plan(args.corrections != "fp16")
            us = ctx.bf_time(50)
            out["kernel1"] = {"us_per_solve": float(np.median(us)), "foci": F, "elements": N, "dtype": "f64", **kernel1_cpu(arr, run_foci[:F])}
            out["end_to_end"] = end_to_end(ol, arr, setup, target, sweep, run_idx[:F], args)
        if args.cpu_seconds > 0 and world == 1:
            pos_m, _, area, _, _ = arr.element_table()
            out["cpu_baseline"] = cpu_baseline(pos_m, area, coords_m, run_foci[0], args.cpu_seconds)
            out["cpu_baseline_c"] = cpu_baseline_c(pos_m, area, coords_m, run_foci[0], min(args.cpu_seconds, 10.0))
        print(json.dumps(out))
        degraded = out["degraded"]
    else:
        degraded = False
    if dist is not None:
        dist.barrier()
        if stuck:            # a thread is still inside a communicator init: no teardown through it -- and the run did not do what was asked
            sys.stdout.flush()
            os._exit(3)
        sf.close()
        dist.destroy_process_group()
    if degraded:
        sys.exit(4)          # (the launcher returns the worst exit code of its ranks)


def config_legs(ol, nat, od, eng, sf, timed, ramp, out_flags):
    """The other BASELINE.json configurations, and arrays the lattice kernels do not serve, as legs of the same driver line -- each planned through
    the product's ShardedField, clock-ramped, timed over its own steps (>= 50), with the kernel the planner chose, its fraction of the 8 TB/s
    roofline on SURVEY 8(d)'s algorithmic bytes, and the error of sampled voxels against the fp64 C oracle measured in this run:
      c2_128         configs[1]  256-element matrix array, single focus, 128^3
      c4_1024x512    configs[3]  1024 elements (32 x 32 @ 1.5 mm), 512^3 @ 0.125 mm, PiecewiseLinear(60, 20) apodization + Direct delays (kernel 1)
      c5_skull_f1/f8 configs[4]  256 elements, 256^3, skull-slab medium (marched ray sums), 1 and 8 foci per launch
      tilted2_*      a two-module TransducerArray on a cylinder (xdc/transducerarray.py:86-115), 2 x 128 elements: tilted normals, no common lattice
      jitter_*       the 16 x 16 array with +-0.1 mm element jitter (seed 147): not a lattice, not mirror-symmetric (1 focus, the 8-focus shard, the 64-focus sweep)
      default_extents_*  the reference's default SimSetup extents (z from -4 mm) at 0.25 mm: launches split at the e4m3 rule's plane cut"""
    from oracle import bf_oracle as bo, c_oracle as co
    from openlifu_amd.engine import grid_from_coords
    from openlifu_amd.seg.seg_methods import skull_slab_volumes
    ctx = eng.ctx
    kinds = {"uniform": nat.APOD_UNIFORM, "maxangle": nat.APOD_MAXANGLE, "piecewise": nat.APOD_PIECEWISE}
    res = {}

    def grid(grid_n, spacing_mm):
        half = (grid_n - 1) / 2 * spacing_mm
        setup = ol.SimSetup(spacing=spacing_mm, x_extent=(-half, half), y_extent=(-half, half), z_extent=(5.0, 5.0 + (grid_n - 1) * spacing_mm))
        origin, spacing, n = grid_from_coords(setup.get_coords())
        return origin, spacing, n, [np.asarray(c.data) * 1e-3 for c in setup.get_coords().values()]

    def run(key, what, arr, g, foci_m, apod=("uniform", 1.0, 0.0), medium=None, steps=200, check=(0,), covered_by=None):
        origin, spacing, n, coords = g
        foci_m = np.atleast_2d(np.asarray(foci_m, dtype=np.float64))
        N, V = arr.numelements(), int(np.prod(n))
        pos_m, _, area, _, _ = arr.element_table()
        ori = np.array([el.orientation for el in arr.elements], dtype=np.float64)
        if medium is None:
            mine = sf.plan_foci_sweep(arr, foci_m, C0, (kinds[apod[0]], apod[1], apod[2]), origin, spacing, n, F0, RHO0, SENS, flags=out_flags)
            foci_run = foci_m[mine]
        else:
            dl, ap = eng.beamform(arr, foci_m, C0)
            sf.plan_slab_sweep(arr, dl, ap, origin, spacing, n, F0, C0, RHO0, SENS, flags=out_flags, medium=medium)
            foci_run = foci_m
        ramp()
        r = timed("none", steps, 10)
        if r is None:
            res[key] = {"what": what, "skipped": timed.last_error}
            return
        e, km = r
        name = ctx.field_variant()
        nf = len(foci_run)
        k_ms = float(np.mean(km))
        bytes_l = (8.0 * V * nf + 32.0 * N * nf) if medium is None else (16.0 * V * nf)     # + 8 B / voxel of medium parameters per focus
        ent = {**mfma_useful(name), "what": what, "kernel": name, "elements": N, "grid": [int(v) for v in n], "foci": nf,
               "dtype": "f32-acc/f16x2+e4m3-corr" if "fp8corr" in name else ("f32-acc/f16x3" if ("field_coset" in name or "field_toep" in name or "field_lattice" in name or "field_mfma" in name) else "f32"),
               "kernel_ms_avg": k_ms, "kernel_launches_timed": int(len(km)), "ms_per_step": e / steps * 1e3, "steps": steps,
               "value": float(V) * N * nf * steps / e / 1e6, "unit": "Mvoxel-elements/s",
               "algorithmic_bytes_per_launch": bytes_l, "roofline_frac": bytes_l / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "traffic": static_traffic(name, int(n[0]))[0]}
        if medium is None:
            ent["parity"] = sampled_parity(ctx, coords, pos_m, area, foci_run, n_samples=8000, check=check, apod=apod, ori=ori, full_volume_test=covered_by)
        else:       # whole z columns against the fp64 marched oracle of the same definition (oracle/field_oracle.c olo_field_columns_hetero_march)
            rng = np.random.default_rng(147)
            cols = np.column_stack([rng.integers(0, n[0], 16), rng.integers(0, n[1], 16)])
            cols[:4] = [[n[0] // 2 - 1, n[1] // 2 - 1], [n[0] // 2, n[1] // 2 + 12], [n[0] // 3, n[1] // 2], [0, 0]]
            sig, ab = co.medium_terms(medium["sound_speed"], medium["attenuation"], C0, F0)
            worst = 0.0
            for f in check:
                if f >= nf:
                    continue
                dl1, ap1 = bo.beamform(pos_m, ori, foci_run[f], C0)
                ref = np.abs(co.field_hetero_march(*coords, sig, ab, pos_m, area, dl1, ap1, F0, C0, SENS, columns=cols))
                got = ctx.field_fetch(f, want=("pmag",))["pmag"][cols[:, 0], cols[:, 1], :]
                worst = max(worst, float(np.abs(got - ref).max() / max(ref.max(), got.max())))
            ent["parity"] = {"max_err_over_column_max": worst, "gate": 1e-5, "columns": int(len(cols)), "foci_checked": [int(f) for f in check if f < nf],
                             "oracle": "oracle/field_oracle.c olo_field_columns_hetero_march (fp64; the build's own definition: parity unpinned)",
                             "covered_by": covered_by}
        res[key] = ent

    focus = np.array([[0.0, 0.0, 40e-3]])
    m16 = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
    run("c2_128", "BASELINE configs[1]: 256-element matrix array, single focus, 128^3 @ 0.5 mm", m16, grid(128, 0.5), focus, steps=500, covered_by="tests/test_gpu_field.py::test_c2_matrix_array_128cubed (full volume)")
    m32 = ol.Transducer.gen_matrix_array(nx=32, ny=32, pitch=1.5, kerf=0.15, units="mm", sensitivity=SENS)
    run("c4_1024x512", "BASELINE configs[3]: 1024-element array (32 x 32 @ 1.5 mm), 512^3 @ 0.125 mm, PiecewiseLinear(zero 60, rolloff 20) apodization + "
        "Direct delays from kernel 1, single focus", m32, grid(512, 0.125), focus, apod=("piecewise", 60.0, 20.0), steps=100,
        covered_by="tests/test_gpu_field.py::test_c4_1024_elements_512cubed_sampled (sampled voxels at 512^3); full volumes of kernel 2f: test_single_column_toeplitz_kernel")
    g256 = grid(256, 0.25)
    wheel = ol.focal_patterns.Wheel(center=True, num_spokes=63, spoke_radius=5.0)
    sweep = np.array([f.get_position(units="m") for f in wheel.get_targets(ol.Point(position=(0, 0, 40), units="mm"))])
    shard = sweep[od.plan_foci_orbits(sweep, 8, centre_xy=(0.0, 0.0))[0]]
    # arrays without a grid-commensurate flat lattice: kernels 2a / 2c
    half = ol.Transducer.gen_matrix_array(nx=8, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
    tilted = ol.TransducerArray.get_concave_cylinder(half, rows=1, cols=2, width=24.0, gap=0.6, roc=80.0, units="mm").to_transducer()
    run("tilted2_f1", "two-module TransducerArray on an 80 mm cylinder (2 x 128 elements, modules tilted -+ 8.8 deg), single focus, 256^3", tilted, g256, focus, covered_by="tests/test_gpu_field.py::test_jittered_tilted_elements_general_variant, test_kernel_families_agree_with_oracle (full volumes, smaller grids)")
    run("tilted2_f8", "the same array, the 8-focus shard", tilted, g256, shard, steps=100, check=(0, 3), covered_by="tests/test_gpu_field.py::test_jittered_tilted_elements_general_variant, test_kernel_families_agree_with_oracle (full volumes, smaller grids)")
    rng = np.random.default_rng(147)
    jit = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=SENS)
    for el in jit.elements:
        el.position = np.asarray(el.position, dtype=np.float64) + rng.uniform(-0.1, 0.1, 3) * np.array([1.0, 1.0, 0.0])
    run("jitter_f1", "16 x 16 array with +-0.1 mm lateral element jitter (seed 147), single focus, 256^3", jit, g256, focus, covered_by="tests/test_gpu_field.py::test_jittered_tilted_elements_general_variant, test_kernel_families_agree_with_oracle (full volumes, smaller grids)")
    run("jitter_f8", "the same array, the 8-focus shard", jit, g256, shard, steps=100, check=(0, 3), covered_by="tests/test_gpu_field.py::test_jittered_tilted_elements_general_variant, test_kernel_families_agree_with_oracle (full volumes, smaller grids)")
    # the reference's DEFAULT SimSetup extents (x, y in +-30 mm, z from -4 mm: through the element plane; sim/sim_setup.py:24-36) at the headline's spacing:
    # the e4m3 rule (include/olx.h) fails in the plane blocks next to the array and the cut comes too late to pay for a second launch, so the planner
    # keeps three fp16 products throughout; the plane count is odd, so the stores are dword-aligned x4 and the XCD group is 17 blocks
    dsetup = ol.SimSetup(spacing=0.25)
    dorigin, dspacing, dn = grid_from_coords(dsetup.get_coords())
    gdef = (dorigin, dspacing, dn, [np.asarray(c.data) * 1e-3 for c in dsetup.get_coords().values()])
    run("default_extents_f8", "the reference's default SimSetup extents at 0.25 mm (241 x 241 x 257, z from -4 mm: the grid passes through the element plane), 16 x 16 array, "
        "the 8-focus shard", m16, gdef, shard, steps=100, check=(0, 3), covered_by="tests/test_gpu_field.py::test_e4m3_rule_near_the_array (full volumes)")
    run("default_extents_f1", "the same grid, single on-axis focus", m16, gdef, focus, steps=200, covered_by="tests/test_gpu_field.py::test_e4m3_rule_near_the_array (full volumes)")
    # ... and the jittered (non-lattice) array on it: kernel 2a's clamp variant, coordinates as (index, residual) -- voxels of the element plane lie a clamp
    # distance (0.033 wavelengths) from elements at lateral coordinates of ~ 8 wavelengths, where absolute fp32 coordinates lost up to 3e-5 of such a term
    run("default_extents_jitter_f1", "the jittered 16 x 16 array on the reference's default SimSetup extents at 0.25 mm, single focus (kernel 2a, split coordinates)", jit, gdef, focus, steps=50,
        covered_by="tests/test_gpu_field.py::test_general_kernels_next_to_the_elements_on_a_wide_grid (full volumes, 0.5 mm)")
    run("jitter_sweep64", "the same array, configs[2]'s whole 64-focus sweep on one GPU (64 steering columns in two launch tiles of 32: kernel 2c's NT = 4 shape)", jit, g256, sweep, steps=20, check=(0, 17, 63), covered_by="tests/test_gpu_field.py::test_many_foci_without_symmetry_uses_wide_mfma_tiles (full volumes, smaller grids)")
    skull = skull_slab_volumes(*g256[3])
    skull["model"] = "marched"
    run("c5_skull_f1", "BASELINE configs[4] on one GPU: 256 elements, 256^3, skull-slab medium, marched ray sums (kernel 2m), one focus per launch",
        m16, g256, focus, medium=skull, steps=50, covered_by="tests/test_gpu_field.py::test_c5_skull_slab_256cubed_marched (whole z columns at 256^3), test_heterogeneous_medium_layered_ray_model (full volumes, small grids)")
    run("c5_skull_f8", "the same medium, the 8-focus shard in one launch sequence (look-ups shared by the foci)", m16, g256, shard, medium=skull, steps=50, check=(0, 3), covered_by="tests/test_gpu_field.py::test_c5_skull_slab_256cubed_marched (whole z columns at 256^3), test_heterogeneous_medium_layered_ray_model (full volumes, small grids)")
    return res


def end_to_end(ol, arr, setup, target, sweep, idx, args):
    """Wall time of the product's API call for the same shard: Protocol.calc_solution(simulate=True, scale=True) --
    beamform, accumulate, scale, aggregate and analyze on the device -- and the device->host bandwidth of materialising
    the per-focus volumes the caller then reads (fresh, caller-owned NumPy arrays)."""
    foci = [sweep[int(i)] for i in idx]
    pattern = ol.focal_patterns.SinglePoint(target_pressure=1e6) if len(foci) == 1 else _ListPattern(ol, foci)
    proto = ol.Protocol(pulse=ol.Pulse(frequency=F0, duration=2e-5), sequence=ol.Sequence(pulse_count=len(foci) * 2, pulse_train_interval=0),
                        focal_pattern=pattern, sim_setup=setup)
    proto.calc_solution(target, arr, simulate=True, scale=True)   # warm (allocations, first touch ...
    ol.get_engine().ctx.field_fetch_all()                         # ... and the fetch workers' pinned staging buffers); nothing kept
    walls = []
    sol = agg = an = None
    for _ in range(7):      # the call is ~10 launches of 0.1 - 0.5 ms each: the first ones after an idle gap run at the idle clock
        sol = agg = an = None   # (drop the previous result first: a live, unread result is rescued to the host -- 1.6 GB -- before its buffers are reused)
        t0 = time.perf_counter()
        sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=True)
        t1 = time.perf_counter()
        walls.append((t1 - t0) * 1e3)
    abytes = sum(np.asarray(agg[k].data).nbytes for k in ("p_min", "p_max", "intensity"))
    ta = time.perf_counter()
    nbytes = 0
    for k in ("p_min", "intensity"):
        nbytes += np.asarray(sol.simulation_result[k].data).nbytes
    t2 = time.perf_counter()
    # The same call when the Dataset factories cannot defer -- what happens with xarray installed, the reference's hard dependency: an
    # xa.Dataset cannot hold a LazyDataArray, so plan/protocol.py fetches the per-focus and the aggregate volumes before it returns.  xarray
    # is absent from this image; the eager stand-ins below (the ones tests/test_gpu_api.py patches in) force that path.
    from openlifu_amd.util import dataset as ds

    class EagerDataArray(ds.DataArray):
        def __init__(self, data, coords=None, dims=None, name=None, attrs=None):
            if isinstance(data, ds.LazyDataArray):
                raise TypeError("cannot defer")
            super().__init__(data, coords=coords, dims=dims, name=name, attrs=attrs)

    class EagerDataset(ds.Dataset):
        def __setitem__(self, name, da):
            if isinstance(da, ds.LazyDataArray):
                raise ValueError("MissingDimensionsError")
            super().__setitem__(name, da)

    class FakeXarray:
        DataArray, Dataset, Coordinates = EagerDataArray, EagerDataset, ds.Coordinates
    mainlobe = [float(v) for v in an.mainlobe_pnp_MPa[:2]]
    sol = agg = an = None
    saved = ds.HAVE_XARRAY, ds._xa
    eager = []
    try:
        ds.HAVE_XARRAY, ds._xa = True, FakeXarray
        for _ in range(4):
            t0 = time.perf_counter()
            sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=True)
            eager.append((time.perf_counter() - t0) * 1e3)
            sol = agg = an = None
    except Exception as e:  # noqa: BLE001 - reported, not fatal
        eager = [float("nan")]
        eager_note = f"eager path failed: {e}"
    else:
        eager_note = ("the same call with the Dataset factories forced eager (what real xarray objects require): the 2 x F per-focus volumes and the "
                      "three aggregate volumes cross PCIe before the call returns")
    finally:
        ds.HAVE_XARRAY, ds._xa = saved
    return {"calc_solution_ms": float(np.median(walls)), "with_xarray_ms": float(np.median(eager[1:] or eager)), "with_xarray_what": eager_note, "calc_solution_ms_all": [round(w, 3) for w in walls], "foci": len(foci), "what": "Protocol.calc_solution(simulate=True, scale=True): "
            "kernel 1 + kernel 2 + device-side scale / aggregate / analyze; aggregate and per-focus volumes left in HBM until read",
            "aggregate_fetch_ms": (ta - t1) * 1e3, "aggregate_fetch_bytes": int(abytes),
            "aggregate_fetch_what": "first .data access of the aggregate Dataset's p_min, p_max and intensity (three fresh NumPy arrays)",
            "fetch_ms": (t2 - ta) * 1e3, "fetch_bytes": int(nbytes), "fetch_GBps": nbytes / max(t2 - ta, 1e-9) / 1e9,
            "fetch_what": "first .data access of simulation_result['p_min'] and ['intensity'] (device -> fresh NumPy arrays)",
            "mainlobe_pnp_MPa": mainlobe}


def _ListPattern(ol, foci):
    """FocalPattern that returns a fixed list of foci (the shard the planner assigned)."""
    from dataclasses import dataclass

    @dataclass
    class ShardPattern(ol.focal_patterns.FocalPattern):
        def get_targets(self, target):
            return [f.copy() for f in foci]

        def num_foci(self):
            return len(foci)
    return ShardPattern(target_pressure=1e6)


if __name__ == "__main__":
    main()
