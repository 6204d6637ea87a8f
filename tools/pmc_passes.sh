#!/bin/bash
# One rocprofv3 --pmc pass per counter group over a target command; per-kernel
# sums and averages by tools/pmc_summary.py.  The program itself follows `--`
# (python3 ...), never a wrapper.
# usage: tools/pmc_passes.sh <tag> <groups: kron|gs|traffic|all> python3 <script> [args]
#   traffic: the three groups that say how many bytes a kernel moved and how its L2 did
#   (FETCH_SIZE, WRITE_SIZE, TCC hits / misses)
# The profiled command must not fork once the profiler's preloaded library has
# initialised the GPU: pass --no-cpu-baseline to bench.py (it also skips the CPU
# baseline by itself when it detects the preload).
set -e
tag=$1; which=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
groups=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_EA0_RDREQ_DRAM_sum"
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
  "SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
  "GRBM_GUI_ACTIVE"
)
[ "$which" = "gs" ] && groups=("${groups[@]:0:6}" "${groups[@]:8:1}")
[ "$which" = "traffic" ] && groups=("${groups[@]:0:3}")
i=0
for grp in "${groups[@]}"; do
  i=$((i+1))
  # a pass that does not come back is cut off and NAMED (its log stays in $out): round 2 lost
  # the name of a hanging group because nothing recorded it (tools/README.md)
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; echo "group failed or timed out: $grp" | tee -a $out/failed_groups.txt; }
  echo "pass $i done: $grp"
done
python3 tools/pmc_summary.py $out > $out/summary.txt
