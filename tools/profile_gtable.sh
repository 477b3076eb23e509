#!/bin/bash
# GPU box: rocprofv3 kernel stats + counter passes of the headline shard with kernel 2g fed from the precomputed geometry table
# (OLX_GTABLE unset / order0) and with the in-kernel generation (OLX_GTABLE=0).  Adds the L2 hit / miss and read-request counters to
# tools/profile_round.sh's groups.   tools/profile_gtable.sh TAG [bench args...]   -> gpurun_out/prof_TAG/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --cpu-seconds 0 --no-extras --steps 300 "$@" > "$out/bench_under_rocprof.json" 2> /dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM TCP_TCC_READ_REQ_sum"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmc$i" -- python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-extras "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py "$out"/pmc* --kernel field_ --json "$out/pmc_summary.json" > /dev/null
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
head -3 "$out/kernel_stats.csv"
cat "$out/pmc_summary.json"
