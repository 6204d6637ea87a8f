#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for jt in 6 3; do for mode in accurate fast; do
timeout -k 10 300 python tools/op_times.py --J_time $jt --J_space 9 --iters 10 --arithmetic $mode 2>&1 | grep -v amdgpu > gpurun_out/r03_final_op_times_J${jt}_${mode}.log || exit 1
echo "J_time=$jt $mode: $(grep -E '^(S|P|Kinv) ' gpurun_out/r03_final_op_times_J${jt}_${mode}.log | tr '\n' ' ')"
done; done
timeout -k 10 600 python tools/op_times.py --J_time 7 --J_space 10 --iters 3 2>&1 | grep -v amdgpu > gpurun_out/r03_final_op_times_config5.log || exit 1
cat gpurun_out/r03_final_op_times_config5.log
