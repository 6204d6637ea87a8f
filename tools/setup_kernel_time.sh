#!/bin/bash
# How long is the DEVICE busy during a set-up?  rocprofv3 --kernel-trace --stats of
# tools/setup_profile.py --timeline (three set-ups of HeatEquationMPI at config 3).
#   tools/setup_kernel_time.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/${tag}_setup_prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_setup_prof -- \
  python3 tools/setup_profile.py --timeline > gpurun_out/${tag}_setup_prof.log 2>&1 || { tail -5 gpurun_out/${tag}_setup_prof.log; exit 1; }
grep "set-up" gpurun_out/${tag}_setup_prof.log
for f in kernel_stats memory_copy_stats; do
  src=$(ls gpurun_out/${tag}_setup_prof/*/*${f}.csv 2>/dev/null | head -1)
  [ -n "$src" ] && cp $src gpurun_out/${tag}_setup_${f}.csv
done
rm -rf gpurun_out/${tag}_setup_prof
python3 - <<PY
import csv
for f in ('kernel_stats', 'memory_copy_stats'):
    try:
        rows = list(csv.DictReader(open('gpurun_out/${tag}_setup_%s.csv' % f)))
    except OSError:
        continue
    total = sum(float(r['TotalDurationNs']) for r in rows) / 1e9
    calls = sum(int(r['Calls']) for r in rows)
    print('%s: %.3f s in %d calls over three set-ups' % (f, total, calls))
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:8]:
        print('   %8.1f ms %6s calls  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], r['Name'][:90]))
PY
