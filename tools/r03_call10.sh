#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "device_plan_construction or multigrid or zero_start or strip_wise or coupling_bands or gauss_seidel or driver_end_to_end or midsize" > gpurun_out/r03_pytest_new_10.log 2>&1
echo "new tests rc=$?"; tail -8 gpurun_out/r03_pytest_new_10.log
timeout -k 10 500 python tools/mg_fill_profile.py --top 28 > gpurun_out/r03_mg_fill_profile_b.log 2>&1; grep "====" gpurun_out/r03_mg_fill_profile_b.log
timeout -k 10 500 python tools/setup_profile.py --top 6 > gpurun_out/r03_setup_profile_b.log 2>&1; grep "====" gpurun_out/r03_setup_profile_b.log
timeout -k 10 300 python - > gpurun_out/r03_setup_wall.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, 'spacetime-fullgrid-parallel_amd')
import torch
torch.zeros(1, device='cuda')
import heateq_mpi as hm
for k in range(3):
    t = time.time()
    h = hm.HeatEquationMPI(J_space=9, J_time=6)
    torch.cuda.synchronize()
    print('HeatEquationMPI(J_time=6, J_space=9) set-up: %.2f s (setup_time %.2f)' % (time.time() - t, h.setup_time), flush=True)
    del h
PY
cat gpurun_out/r03_setup_wall.log
