#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "without_dictionary or sharing_one_gpu or long_enough_for_row_pairs" > gpurun_out/r03_pytest_new_8.log 2>&1
echo "new tests rc=$?"; tail -8 gpurun_out/r03_pytest_new_8.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_8.log 2>&1
echo "rc=$?"
tail -8 gpurun_out/r03_pytest_gpu_8.log
