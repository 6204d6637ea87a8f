#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "explicit_value or kron_pack_row_pairs" > gpurun_out/r03_pytest_new_5.log 2>&1
echo "new tests rc=$?"; tail -4 gpurun_out/r03_pytest_new_5.log
timeout -k 10 600 python tools/kron_ab.py --problem lshape_jitter --J_space 8 --n_loc 33 --variants "plain;pack" > gpurun_out/r03_ab_jitter_J8_33.log 2>&1
echo "ab1 rc=$?"; tail -3 gpurun_out/r03_ab_jitter_J8_33.log
timeout -k 10 600 python tools/kron_ab.py --problem lshape_jitter --J_space 8 --n_loc 65 --variants "plain;pack" > gpurun_out/r03_ab_jitter_J8_65.log 2>&1
echo "ab2 rc=$?"; tail -3 gpurun_out/r03_ab_jitter_J8_65.log
timeout -k 10 600 python tools/setup_profile.py --top 10 > gpurun_out/r03_setup_profile.log 2>&1
echo "setup rc=$?"; grep "====" gpurun_out/r03_setup_profile.log
timeout -k 10 600 python tools/op_times.py --J_time 3 --J_space 9 > gpurun_out/r03_op_times_J3_J9.log 2>&1
echo "op rc=$?"; cat gpurun_out/r03_op_times_J3_J9.log | tail -9
timeout -k 10 600 python tools/op_times.py --J_time 6 --J_space 9 > gpurun_out/r03_op_times_J6_J9.log 2>&1
echo "op rc=$?"; cat gpurun_out/r03_op_times_J6_J9.log | tail -9
