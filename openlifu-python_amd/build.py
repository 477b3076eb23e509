#!/usr/bin/env python3
"""Build libolx.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

  python openlifu-python_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so lands in openlifu-python_amd/lib/ (git-ignored,
travels with gpurun snapshots)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "olx.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "olx_kernels.hip.h"), os.path.join(HERE, "..", "include", "olx.h")]
OUT = os.path.join(HERE, "lib", "libolx.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
         "-I/opt/rocm/include"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm 7.x expected under /opt/rocm)")


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc()] + FLAGS + ["-o", OUT] + SRC + ["-ldl"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
