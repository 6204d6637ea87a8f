#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "explicit_value or c_abi_communication or direct_preconditioner or serial_driver" > gpurun_out/r03_pytest_new_4.log 2>&1
echo "new tests rc=$?"; tail -8 gpurun_out/r03_pytest_new_4.log
timeout -k 10 600 python tools/kron_ab.py --problem lshape_jitter --J_space 8 --n_loc 33 --variants "plain;pack" > gpurun_out/r03_ab_jitter_J8_33.log 2>&1
echo "ab1 rc=$?"; tail -4 gpurun_out/r03_ab_jitter_J8_33.log
timeout -k 10 600 python tools/kron_ab.py --problem lshape_jitter --J_space 8 --n_loc 65 --variants "plain;pack" > gpurun_out/r03_ab_jitter_J8_65.log 2>&1
echo "ab2 rc=$?"; tail -4 gpurun_out/r03_ab_jitter_J8_65.log
timeout -k 10 600 python tools/kron_ab.py --problem square --J_space 9 --n_loc 65 --variants "plain;pack1;pack" > gpurun_out/r03_ab_square_J9_65.log 2>&1
echo "ab3 rc=$?"; tail -5 gpurun_out/r03_ab_square_J9_65.log
