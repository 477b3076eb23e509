#!/bin/bash
# Developer tool (GPU box): several rocprofv3 --pmc passes over `python bench.py <args>`, 8 counters per pass.
#   tools/pmc_run.sh OUTDIR [bench args...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
 "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_IFETCH" \
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES" ; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -- python bench.py --cpu-seconds 0 --steps 5 --warmup 1 "$@" > /dev/null 2>&1
  i=$((i+1))
done
python tools/pmc_summary.py "$out"/p* --kernel field_ --json "$out/summary.json"
