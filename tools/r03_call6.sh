#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "recorded_vcycles or explicit_value or kron_pack_row_pairs or strip_wise or coarse_subcycle" > gpurun_out/r03_pytest_new_6.log 2>&1
echo "new tests rc=$?"; tail -6 gpurun_out/r03_pytest_new_6.log
for g in 0 1; do
for jt in 3 4 6; do
timeout -k 10 600 python tools/op_times.py --J_time $jt --J_space 9 --iters 10 --tune mg_graph=$g > gpurun_out/r03_op_graph${g}_J${jt}_J9.log 2>&1
echo "graph=$g J_time=$jt rc=$?"; grep -E "^(S|P|Kinv) " gpurun_out/r03_op_graph${g}_J${jt}_J9.log
done
done
