#!/bin/bash
# GPU box: produce the per-round profile artefacts for one bench configuration.
#   tools/profile_round.sh TAG [bench args...]      (writes gpurun_out/prof_TAG/{stats,pmc*}, summary json)
# Kernel stats: rocprofv3 --kernel-trace --stats over the SAME command line as the bench (default steps / warm-up).  Counters: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass; MI355X_MICROARCH.md, rocprofv3 PMC slots).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --cpu-seconds 0 "$@" > "$out/bench_under_rocprof.json" 2> /dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmc$i" -- python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py "$out"/pmc* --kernel field_ --json "$out/pmc_summary.json" > /dev/null
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
python3 - "$out" "$@" <<'PY'
import json, sys
out = sys.argv[1]
s = json.load(open(out + "/pmc_summary.json"))
for k, v in s.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        v["hbm_note"] = "FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)"
try:
    s["library_built_from_commit"] = open("openlifu-python_amd/lib/libolx.so.stamp").read().strip()
except OSError:
    s["library_built_from_commit"] = "unknown"
s["command"] = "tools/profile_round.sh: rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 " + " ".join(sys.argv[2:])
json.dump(s, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in s.items() if isinstance(v, dict)}))
PY
cp openlifu-python_amd/lib/libolx.so.stamp "$out/kernel_stats.commit" 2>/dev/null
head -4 "$out/kernel_stats.csv"
