#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "strip_wise or coupling_bands or two_stream or midsize or plan_from_csr" > gpurun_out/r03_pytest_new_26.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03_pytest_new_26.log
for rep in 1 2; do
timeout -k 10 300 python tools/op_times.py --J_time 6 --J_space 9 --iters 10 > gpurun_out/r03_strip_pct_$rep.log 2>&1
echo "per-plan strips: $(grep -E '^(S|P|Kinv) ' gpurun_out/r03_strip_pct_$rep.log | tr '\n' ' ')"
done
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_strip_pct.json 2> gpurun_out/r03_bench_strip_pct.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/r03_bench_strip_pct.json'))
print('bench: %.4f ms/step frac %.3f; pcg %.2f it/s %.2f ms/iter' % (b['ms_per_step'], b['roofline']['frac'], b['pcg']['iters_per_s'], b['pcg']['ms_per_iter']))
PY
