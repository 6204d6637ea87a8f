#!/bin/bash
# First set-up of a process under the allocator policy: heap without huge pages
# (STK_HEAP_HUGE_GB=0), with the policy's own advice (default: the next 3 GB of the heap),
# and with glibc's tunable glibc.malloc.hugetlb=1 (process start only) for comparison.
cd "$GRAFT_REPO_ROOT" || exit 1
ldd --version | head -1
for v in none advised tunable none advised tunable; do
  log=gpurun_out/$1_thp_${v}_$RANDOM.log
  case $v in
    none) STK_HEAP_HUGE_GB=0 python tools/setup_faults.py > $log 2>&1 || { tail -3 $log; exit 1; } ;;
    advised) python tools/setup_faults.py > $log 2>&1 || { tail -3 $log; exit 1; } ;;
    tunable) STK_HEAP_HUGE_GB=0 GLIBC_TUNABLES=glibc.malloc.hugetlb=1 python tools/setup_faults.py > $log 2>&1 || { tail -3 $log; exit 1; } ;;
  esac
  echo "$v: $(grep '^set-up' $log | cut -c1-95 | tr '\n' '|')"
done
