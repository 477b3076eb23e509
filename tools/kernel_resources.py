#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of the device code (cross-compiles for gfx950, no GPU needed).

  python tools/kernel_resources.py [name-filter]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
import glob
out = ""
for src in sorted(glob.glob(os.path.join(HERE, "openlifu-python_amd", "csrc", "*.hip"))):      # every translation unit (one per kernel family)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-c", "-o", "/tmp/olx_dev.o", src,
           "-I/opt/rocm/include", "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[2:] if a.startswith("-")]
    if flt and flt not in open(src).read():
        continue
    out += subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?): +(\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k.endswith("Name"):
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.check_output(["c++filt", r["name"]], text=True).strip()
    name = re.sub(r"\(.*", "", name).replace("void olx::", "")
    if flt not in name:
        continue
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{name:60s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>4s} spill {g('VGPRs Spill'):>3s} scratch {g('ScratchSize [bytes/lane]'):>4s} "
          f"lds {g('LDS Size [bytes/block]'):>6s} occ {g('Occupancy [waves/SIMD]')}")
