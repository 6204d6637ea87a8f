#!/bin/bash
# Set-up timeline under glibc allocator settings: do the planner threads serialise on
# the address-space lock (every large NumPy temporary is an mmap / page faults / munmap,
# every pageable upload pins its pages)?
#   tools/setup_malloc_ab.sh <tag>
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  name=$1; shift
  log=gpurun_out/${tag}_malloc_${name}.log
  env "$@" python tools/setup_profile.py --timeline > $log 2>&1 || { tail -5 $log; exit 1; }
  echo "$name: $(grep '^set-up' $log | tr '\n' ' ')"
}
run untouched STK_KEEP_MALLOC=1
run env_heap_only STK_KEEP_MALLOC=1 MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=100000000000 MALLOC_ARENA_MAX=1
run env_heap_arenas STK_KEEP_MALLOC=1 MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=100000000000
run scoped STK_NOTHING=1
run scoped_early_arena STK_EARLY_ARENA=1
run keep_to_the_heap STK_HEAP=1
run untouched2 STK_KEEP_MALLOC=1
