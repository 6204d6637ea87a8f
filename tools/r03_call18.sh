#!/bin/bash
# round 3 final: profile round (bench, rocprof stats, PMC), operator times, smoke
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_round.sh r03_final > gpurun_out/r03_final_profile_round.log 2>&1; echo "profile rc=$?"; tail -12 gpurun_out/r03_final_profile_round.log
timeout -k 10 400 python tools/op_times.py --J_time 6 --J_space 9 --iters 10 > gpurun_out/r03_final/op_times.log 2>&1; grep -E "^(W|WT|S|P|Kinv|A_x|axpy|dot) " gpurun_out/r03_final/op_times.log
timeout -k 10 400 python tools/op_times.py --J_time 3 --J_space 9 --iters 10 > gpurun_out/r03_final/op_times_J3.log 2>&1; grep -E "^(S|P|Kinv) " gpurun_out/r03_final/op_times_J3.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final/smoke.log 2>&1; tail -1 gpurun_out/r03_final/smoke.log
