#!/bin/bash
# GPU box: CU clock, busy cycles and issue counts of the headline launch for several builds of the library (one rocprofv3 --pmc pass each):
#   tools/clock_under_load.sh LIB [LIB ...]      (names under openlifu-python_amd/lib/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for l in "$@"; do
  out=gpurun_out/clk_${l%.so}
  rm -rf "$out"
  OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$l rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d "$out" -- python3 bench.py --cpu-seconds 0 --no-extras --steps 200 --warmup 30 > /dev/null 2>&1
  python3 tools/pmc_summary.py "$out" --kernel field_ --json "$out/summary.json" > /dev/null
  python3 - "$out/summary.json" "$l" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and "SQ_BUSY_CU_CYCLES" in v:
        us = v["avg_us_under_pmc"]; cyc = v["SQ_BUSY_CU_CYCLES"] / 256
        print(f"{sys.argv[2]:24s} {us:8.1f} us   {cyc / 1e3:8.1f} k cycles per CU   clock {cyc / us / 1e3:5.3f} GHz   matrix pipe busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.3f}")
PY
done
