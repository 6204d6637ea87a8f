#!/bin/bash
# round 3, GPU call 1: synchronisation sizing prototype, history attribution, test suite
set -o pipefail
mkdir -p gpurun_out
for w in 1 2 3; do
  timeout -k 10 120 tools/bin/gs_sync_proto $w 2000 3072 60 >> gpurun_out/r03_gs_sync_proto.log 2>&1 || echo "proto $w failed rc=$?" >> gpurun_out/r03_gs_sync_proto.log
done
timeout -k 10 120 tools/bin/gs_sync_proto 2 2000 3072 240 >> gpurun_out/r03_gs_sync_proto.log 2>&1
timeout -k 10 900 python tools/history_attribution.py --out gpurun_out/r03_history_attribution.json > gpurun_out/r03_history_attribution.log 2>&1 &&
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu_1.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/r03_pytest_gpu_1.log
