#!/bin/bash
# GPU box: counters of the store path (vector memory unit, L1, L2, memory interface) for one bench configuration -- one rocprofv3 --pmc pass per group;
# a group with a counter this part does not have is reported and skipped.
# (TCC_WRITEBACK_sum / TCC_BUSY_sum hang rocprofv3 on this pool: not collected.)   tools/pmc_store_path.sh OUTDIR [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in \
 "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_BUSY_CU_CYCLES" \
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" \
 "TCC_REQ_sum TCC_WRITE_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
 "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" ; do
  if rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -- python3 bench.py --cpu-seconds 0 --no-extras --steps 30 --warmup 5 "$@" > /dev/null 2> "$out/p$i.err"; then :; else echo "group $i failed: $set"; tail -2 "$out/p$i.err"; fi
  i=$((i+1))
done
python3 tools/pmc_summary.py "$out"/p? --kernel field_ --json "$out/summary.json"
