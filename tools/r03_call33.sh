#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "oracle_trajectory or full_size or driver or restricted_residual or distributed" > gpurun_out/r03_pytest_33.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r03_pytest_33.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_33.json 2> gpurun_out/r03_bench_33.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/r03_bench_33.json'))
print('bench: %.4f ms/step frac %.3f' % (b['ms_per_step'], b['roofline']['frac']))
for k in ('pcg','pcg_fast'):
    p=b[k]; print(k, p['arithmetic'], '%.2f it/s %.2f ms/iter setup %.2f s' % (p['iters_per_s'], p['ms_per_iter'], p['setup_s']))
PY
python -c "
import json; d=json.load(open('gpurun_out/parity_history_dev.json'))
for k,v in sorted(d.items()):
    if 'accurate' in k: print(k, '%.2e' % v['max_rel_dev_r_dot_Pr'])
"
