#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for mb in 250 120 180 400; do
timeout -k 10 300 python tools/op_times.py --J_time 6 --J_space 9 --iters 10 --tune mg_strip_mb=$mb > gpurun_out/r03_strip_${mb}.log 2>&1
echo "strip_mb=$mb: $(grep -E '^(S|P|Kinv) ' gpurun_out/r03_strip_${mb}.log | tr '\n' ' ')"
done
