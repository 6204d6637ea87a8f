#!/bin/bash
# Collects HBM / L2 / SQ counters for the bench's dominant kernel, one rocprofv3
# pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/pmc_kron.sh <tag> [bench args]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --solve-iters 0 --no-cpu-baseline "$@" > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out
