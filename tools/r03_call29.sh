#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r03_short_slab
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/op_times.py --J_time 3 --J_space 9 --iters 20 > $out/op_times.log 2> $out/prof.log || { tail -5 $out/prof.log; exit 1; }
cp $(ls $out/prof/*/*kernel_stats.csv | head -1) gpurun_out/r03_short_slab_kernel_stats.csv
rm -rf $out/prof
cat $out/op_times.log
head -25 gpurun_out/r03_short_slab_kernel_stats.csv | cut -c1-200
