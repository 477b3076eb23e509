#!/bin/bash
# GPU box: the round-3 profile set (kernel stats + PMC summaries per bench shape, scan kernels).  Results under gpurun_out/prof_r03_*.
#   tools/profile_r03.sh [shape ...]      shapes: fp16 fp8 f1 offaxis asym f64 scans (default: all)
cd "$GRAFT_REPO_ROOT" || exit 1
shapes=${@:-fp16 fp8 f1 offaxis asym f64 scans}
for s in $shapes; do
  case $s in
    fp16)    bash tools/profile_round.sh r03_cosetp_f8_fp16 --no-extras ;;
    fp8)     bash tools/profile_round.sh r03_cosetp_f8_fp8 --no-extras --corrections fp8 ;;
    f1)      bash tools/profile_round.sh r03_toep_f1 --no-extras --foci-per-gpu 1 ;;
    offaxis) bash tools/profile_round.sh r03_offaxis_f1 --no-extras --foci-per-gpu 1 --offset-mm 1.3,0.7 ;;
    asym)    bash tools/profile_round.sh r03_asym_f8 --no-extras --offset-mm 1.3,0.7 ;;
    f64)     bash tools/profile_round.sh r03_coset_f64 --no-extras --foci-per-gpu 64 --steps 100 --warmup 10 ;;
    scans)
      out=gpurun_out/prof_r03_scans; mkdir -p $out
      cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
      rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/scan_bench.py > $out/scan_bench.txt 2>/dev/null
      find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
      for set in "FETCH_SIZE" "WRITE_SIZE"; do
        rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$set -- python3 tools/scan_bench.py > /dev/null 2>&1
      done
      python3 tools/pmc_summary.py $out/pmc_* --kernel field_ --json $out/pmc_summary.json > /dev/null
      python3 tools/pmc_summary.py $out/pmc_* --kernel offset_grid --json $out/pmc_summary_og.json > /dev/null
      cat $out/scan_bench.txt ;;
  esac
done
