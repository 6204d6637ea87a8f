#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "two_stream or heat_operators_and_solve or driver_end_to_end or sharing_one_gpu or trajectory or every_operator_class or direct_preconditioner" > gpurun_out/r03_pytest_new_14.log 2>&1
echo "new tests rc=$?"; tail -6 gpurun_out/r03_pytest_new_14.log
timeout -k 10 400 python tools/op_times.py --J_time 6 --J_space 9 --iters 10 > gpurun_out/r03_op_times_two_streams_J6.log 2>&1; grep -E "^(S|P|Kinv) " gpurun_out/r03_op_times_two_streams_J6.log
timeout -k 10 400 python tools/op_times.py --J_time 3 --J_space 9 --iters 10 > gpurun_out/r03_op_times_two_streams_J3.log 2>&1; grep -E "^(S|P|Kinv) " gpurun_out/r03_op_times_two_streams_J3.log
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_two_streams.json 2> gpurun_out/r03_bench_two_streams.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/r03_bench_two_streams.json'))
print('bench: %.4f ms/step frac %.3f; pcg %.2f it/s %.2f ms/iter' % (b['ms_per_step'], b['roofline']['frac'], b['pcg']['iters_per_s'], b['pcg']['ms_per_iter']))
PY
