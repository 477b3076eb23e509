#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean of every counter over its dispatches.

  python tools/pmc_summary.py gpurun_out/pmc_dir [more dirs...] [--kernel substr] [--json out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
kfilter = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else ""
jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if kfilter in args:
    args.remove(kfilter)
if jout in args:
    args.remove(jout)
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void olx::", "")
            if kfilter not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            key = (name, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
out = {}
for name, ctrs in acc.items():
    out[name] = {"dispatches": len(dur[name]), "avg_us_under_pmc": sum(dur[name]) / max(len(dur[name]), 1)}
    for c, v in sorted(ctrs.items()):
        out[name][c] = sum(v) / len(v)
    print(name)
    for k, v in out[name].items():
        print(f"    {k:32s} {v:16.1f}")
if jout:
    json.dump(out, open(jout, "w"), indent=1)
