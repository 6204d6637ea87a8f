#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_gs_tile_sizes.log
for rpt in 64 256 1024 2048 8192; do
echo "== STK_ROWS_PER_TILE=$rpt" >> gpurun_out/r03_gs_tile_sizes.log
STK_ROWS_PER_TILE=$rpt timeout -k 10 200 python tools/gs_sweep_time.py 9 65 2>&1 | grep J_space >> gpurun_out/r03_gs_tile_sizes.log || exit 1
done
cat gpurun_out/r03_gs_tile_sizes.log
for rpt in 256 2048; do
echo "== STK_ROWS_PER_TILE=$rpt op_times" >> gpurun_out/r03_gs_tile_sizes.log
STK_ROWS_PER_TILE=$rpt timeout -k 10 300 python tools/op_times.py --J_time 6 --J_space 9 --iters 10 2>&1 | grep -E "^(S|P|Kinv) " | tee -a gpurun_out/r03_gs_tile_sizes.log || exit 1
done
