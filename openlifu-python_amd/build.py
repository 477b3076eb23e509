#!/usr/bin/env python3
"""Build libolx.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

  python openlifu-python_amd/build.py [--force] [-DOLX_EXP_...] [--out lib/libolx_exp.so]

hipcc cross-compiles without a GPU.  Every csrc/*.hip is one translation unit (the host side olx.hip plus
one k_*.hip per kernel-2 family), compiled in parallel to build/*.o and linked into
openlifu-python_amd/lib/libolx.so (git-ignored, travels with gpurun snapshots).  A unit is rebuilt when it or any
header it could include (csrc/*.h, include/olx.h) is newer than its object."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libolx.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I/opt/rocm/include"]
# per-unit additions.  k_hmarch.hip: the SLP vectoriser pairs the scalar lerps of the one-sum look-up into v_pk_* with three v_mov shuffles per
# look-up (its explicit float2 arithmetic of the two-sum form stays packed either way)
UNIT_FLAGS = {"k_hmarch.hip": ["-fno-slp-vectorize"]}


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


def build(force: bool = False, defines=(), out: str = OUT) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))     # (*.cpp: host-only units, e.g. olx_plan.cpp)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "olx.h")]
    tag = "" if not defines else "_" + "_".join(d.lstrip("-D").replace("=", "") for d in defines)
    if defines:     # every non-product build (debug library, timing / tracing builds) also honours the developer pins the product ignores:
        defines = list(defines) + ["-DOLX_DEV_PINS"]      # OLX_FP8_CORRECTION=1 (forces e4m3 past its error rule), OLX_EXP_KGRP, OLX_EXP_TOEP_SAW
    objdir = os.path.join(HERE, "build" + tag)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    cc = hipcc()
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            host_only = s.endswith(".cpp")
            jobs.append([cc] + [f for f in FLAGS if not (host_only and f.startswith("--offload-arch"))] + list(defines) + UNIT_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for rc, cmd in zip(ex.map(lambda c: subprocess.run(c).returncode, jobs), jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed: " + " ".join(cmd))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(o) for o in objs):
        subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    write_stamp(out)
    return out


def write_stamp(out: str) -> None:
    """<library>.stamp = the commit the library was built from (+ "-dirty" when the csrc / include tree differs from it): it travels with the
    library to the GPU box (which has no .git), so that bench.py and tools/profile_round.sh can name the code their numbers belong to."""
    root = os.path.join(HERE, "..")
    try:
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
        dirty = subprocess.run(["git", "-C", root, "diff", "--quiet", "HEAD", "--", "openlifu-python_amd/csrc", "include"]).returncode != 0
        with open(out + ".stamp", "w") as f:
            f.write(head + ("-dirty" if dirty else "") + "\n")
    except Exception:  # noqa: BLE001 - no git here (the GPU box): keep the stamp that travelled with the library
        pass


if __name__ == "__main__":
    args = sys.argv[1:]
    out = OUT
    if "--out" in args:
        out = os.path.join(HERE, args[args.index("--out") + 1])
    print(build(force="--force" in args, defines=[a for a in args if a.startswith("-D")], out=out))
