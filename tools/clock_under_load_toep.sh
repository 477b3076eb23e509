#!/bin/bash
# GPU box: CU clock and busy cycles of kernel 2f on BASELINE configs[3] and on the single focus at 256^3 for several builds of the library (the phase-skip builds of
# tools/exp/toep_phase_skip.patch: does the chip hold a different clock under the table generation, the contraction and the epilogue alone?):
#   tools/clock_under_load_toep.sh LIB [LIB ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in c4 f1; do
  if [ $shape = c4 ]; then args="--foci-per-gpu 1 --steps 60 --warmup 10 --elements 32x32 --pitch-mm 1.5 --grid 512 --spacing-mm 0.125"; else args="--foci-per-gpu 1 --steps 400 --warmup 50"; fi
  for l in "$@"; do
    out=gpurun_out/clkt_${shape}_${l%.so}
    rm -rf "$out"
    OLX_LIB_PATH=$GRAFT_REPO_ROOT/openlifu-python_amd/lib/$l rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out" -- python3 bench.py --cpu-seconds 0 --no-extras $args > /dev/null 2>&1
    python3 tools/pmc_summary.py "$out" --kernel field_toep --json "$out/summary.json" > /dev/null
    python3 - "$out/summary.json" "$shape $l" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and "SQ_BUSY_CU_CYCLES" in v:
        us = v["avg_us_under_pmc"]; cyc = v["SQ_BUSY_CU_CYCLES"] / 256
        print(f"{sys.argv[2]:28s} {us:8.1f} us   {cyc / 1e3:8.1f} k busy cycles per CU   clock >= {cyc / us / 1e3:5.3f} GHz (busy cycles / duration)   matrix pipe busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.3f}")
PY
  done
done
