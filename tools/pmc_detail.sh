#!/bin/bash
# Detailed SQ / TA / TCP counters for the bench's kernels (one pass per group).
# usage: tools/pmc_detail.sh <tag> [bench args]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcd_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --solve-iters 0 --no-cpu-baseline "$@" > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; echo "group failed: $grp"; }
done
python3 tools/pmc_summary.py $out
