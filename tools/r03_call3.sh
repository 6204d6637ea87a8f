#!/bin/bash
# round 3, GPU call 3: explicit-value pairs (test + A/B), trajectory tests in both arithmetic modes, full suite
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/parity_history_dev.json
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "explicit_value or byte_moving or kron_pack_row_pairs" > gpurun_out/r03_pytest_new_3.log 2>&1
echo "new tests rc=$?"; tail -8 gpurun_out/r03_pytest_new_3.log
timeout -k 10 600 python tools/kron_ab.py --problem lshape_jitter --J_space 8 --n_loc 33 --variants "plain;pack" > gpurun_out/r03_ab_jitter_J8_33.log 2>&1
echo "ab1 rc=$?"; tail -4 gpurun_out/r03_ab_jitter_J8_33.log
timeout -k 10 600 python tools/kron_ab.py --problem lshape_jitter --J_space 9 --n_loc 65 --variants "plain;pack" > gpurun_out/r03_ab_jitter_J9_65.log 2>&1
echo "ab2 rc=$?"; tail -4 gpurun_out/r03_ab_jitter_J9_65.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_3.log 2>&1
echo "rc=$?"
tail -30 gpurun_out/r03_pytest_gpu_3.log
