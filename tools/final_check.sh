#!/bin/bash
# What the driver runs at round end, in one GPU call: the GPU suite, then smoke().
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/$1_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/$1_pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$1_smoke.log 2>&1
rc=$?; tail -2 gpurun_out/$1_smoke.log; exit $rc
