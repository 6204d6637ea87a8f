#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 tools/bin/gs_tile_proto 1023 65 66 28 8 5 > gpurun_out/r03_gs_tile_proto_ld66.log 2>&1 && cat gpurun_out/r03_gs_tile_proto_ld66.log &&
timeout -k 10 120 tools/bin/gs_tile_proto 1023 64 64 32 8 5 > gpurun_out/r03_gs_tile_proto_ld64.log 2>&1 && cat gpurun_out/r03_gs_tile_proto_ld64.log &&
timeout -k 10 120 tools/bin/gs_tile_proto 1023 9 10 28 8 5 > gpurun_out/r03_gs_tile_proto_ld10.log 2>&1 && cat gpurun_out/r03_gs_tile_proto_ld10.log &&
timeout -k 10 300 python tools/kron_ab.py --variants "pack;pack,pack_alternate=1" > gpurun_out/r03_ab_alternate.log 2>&1 && grep -E "^pack" gpurun_out/r03_ab_alternate.log &&
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "oracle_trajectory and accurate" > gpurun_out/r03_pytest_accurate.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_pytest_accurate.log
