#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/setup_profile.py --timeline > gpurun_out/r03_setup_timeline.log 2>&1; cat gpurun_out/r03_setup_timeline.log | tail -22
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_12.log 2>&1
echo "rc=$?"
tail -8 gpurun_out/r03_pytest_gpu_12.log
