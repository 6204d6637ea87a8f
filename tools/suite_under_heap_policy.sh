#!/bin/bash
# The GPU suite under the allocator policy the drivers and bench.py run with
# (tests/conftest.py: STK_TEST_HEAP=1 -> source.host_malloc.keep_to_the_heap()).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
STK_TEST_HEAP=1 timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/$1_pytest_heap.log 2>&1
rc=$?; tail -4 gpurun_out/$1_pytest_heap.log; exit $rc
