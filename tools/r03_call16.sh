#!/bin/bash
# final of round 3: full GPU suite, profile round, operator times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_16.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03_pytest_gpu_16.log
