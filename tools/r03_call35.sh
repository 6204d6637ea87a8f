#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "coarse_subcycle or multigrid_match or random_algebraic or zero_start or oracle_trajectory" 2>&1 | tail -3 || exit 1
: > gpurun_out/r03_coarse_ahead.log
for jt in 3 6; do for ah in 0 1 0 1; do
echo "J_time=$jt mg_coarse_ahead=$ah: $(timeout -k 10 300 python tools/op_times.py --J_time $jt --J_space 9 --iters 20 --arithmetic fast --tune mg_coarse_ahead=$ah 2>&1 | grep -E '^(S|P|Kinv) ' | tr '\n' ' ')" | tee -a gpurun_out/r03_coarse_ahead.log
done; done
