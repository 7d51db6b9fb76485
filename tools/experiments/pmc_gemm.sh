#!/bin/bash
# usage: tools/experiments/pmc_gemm.sh OUTDIR  (on the GPU box): one rocprofv3 run per counter group (--pmc with --kernel-trace only)
set -e
OUT=$1; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $REPO/tools/experiments/pmc_gemm.py run > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $grp"; tail -3 $OUT/p$i.log; }
  echo "pass $i done: $grp"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCC_HIT_sum TCC_MISS_sum
GROUPS
