# GPU box: per-launch durations of kernel 2m's launch classes on configs[4] (one focus): look-up launches (ES = 1) and writer launches (ES = 16).
#   tools/exp_hmarch_mall.sh [elements ...]      default 16x16; round 5 also ran 8x16 and 8x8 to see whether a U double buffer that fits the
#   256 MB Infinity Cache makes the writers faster per element (it does not: 72.8 / 34.2 / 23.3 us per launch at 256 / 128 / 64 elements)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for el in ${@:-16x16}; do
  rm -rf gpurun_out/exp_mall_$el
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_mall_$el -- python3 bench.py --medium skull --foci-per-gpu 1 --elements $el --no-extras --cpu-seconds 0 --steps 20 --warmup 3 > /dev/null 2>&1
  f=$(find gpurun_out/exp_mall_$el -name "*kernel_stats.csv" | head -1)
  echo "== $el"
  python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "hmarch" in r["Name"]:
        t = re.search(r"field_hmarch_k<([^>]*)>", r["Name"]).group(1)
        print(t, "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
done
