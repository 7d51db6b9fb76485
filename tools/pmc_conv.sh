#!/bin/bash
# usage: tools/pmc_conv.sh OUTDIR   (on the GPU box; env UMI_CONV3X3_IMPL selects the kernel)
# one rocprofv3 run per counter group (--pmc with --kernel-trace only)
set -e
OUT=$1; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $REPO/tools/pmc_conv.py run > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $grp"; tail -3 $OUT/p$i.log; }
  echo "pass $i done: $grp"
done < <(if [ -n "$PMC_GROUPS" ]; then echo "$PMC_GROUPS" | tr ';' '\n'; else cat <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_EA0_RDREQ_sum
GROUPS
fi)
