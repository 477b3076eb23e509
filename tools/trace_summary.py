#!/usr/bin/env python3
"""Per (kernel, grid size) launch statistics from a rocprofv3 --kernel-trace CSV directory:
   python tools/trace_summary.py DIR [name-filter]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if flt in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0][-48:]
        agg[(name, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("VGPR_Count"), r.get("LDS_Block_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:50s} grid {k[1]:>9s} vgpr {k[2]} lds {k[3]} calls {len(v):5d} avg {sum(v)/len(v):9.1f} us  med {v[len(v)//2]:9.1f}  total {sum(v)/1e3:8.2f} ms")
