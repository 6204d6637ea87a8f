#!/bin/bash
# round 3, GPU call 2: new tests first, history attribution with the restricted-residual switch, slab shapes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "slab_storage or byte_moving or direct_inverse_above or kron_pack_row_pairs or spawns or test_every_operator_class or sharing_one_gpu or row_pairs" > gpurun_out/r03_pytest_new_2.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/r03_pytest_new_2.log
timeout -k 10 900 python tools/history_attribution.py --out gpurun_out/r03_history_attribution_b.json > gpurun_out/r03_history_attribution_b.log 2>&1
echo "attribution rc=$?"
timeout -k 10 600 python tools/kron_slab_shapes.py --flags 3 > gpurun_out/r03_slab_shapes.log 2>&1
echo "slab shapes rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu_2.log 2>&1
echo "rc=$?"
tail -5 gpurun_out/r03_pytest_gpu_2.log
