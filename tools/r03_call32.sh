#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_32.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r03_pytest_gpu_32.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -2
