#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/setup_profile.py --timeline > gpurun_out/r03_setup_timeline_b.log 2>&1 && cat gpurun_out/r03_setup_timeline_b.log &&
timeout -k 10 400 python tools/setup_profile.py --top 10 > gpurun_out/r03_setup_profile_c.log 2>&1 && grep "====" gpurun_out/r03_setup_profile_c.log &&
timeout -k 10 300 python tools/gs_sweep_time.py 9 65 > gpurun_out/r03_gs_sweep_65.log 2>&1 && cat gpurun_out/r03_gs_sweep_65.log &&
timeout -k 10 300 python tools/gs_sweep_time.py 9 9 > gpurun_out/r03_gs_sweep_9.log 2>&1 && cat gpurun_out/r03_gs_sweep_9.log &&
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "(oracle_trajectory and fast) or driver or smoke or serial" > gpurun_out/r03_pytest_28.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_pytest_28.log
