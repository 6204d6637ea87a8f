#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for jt in 3 6; do
for cr in 4096 1024 256 16; do
timeout -k 10 300 python tools/op_times.py --J_time $jt --J_space 9 --iters 10 --coarse-rows $cr > gpurun_out/r03_coarse_${jt}_${cr}.log 2>&1
echo "J_time=$jt coarse_rows=$cr: $(grep -E '^(S|P|Kinv) ' gpurun_out/r03_coarse_${jt}_${cr}.log | tr '\n' ' ')"
done
done
