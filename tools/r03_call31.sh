#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_baseline_configs_full_size_against_oracle > gpurun_out/r03_pytest_gpu_31.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/r03_pytest_gpu_31.log
[ $rc -eq 0 ] || exit $rc
