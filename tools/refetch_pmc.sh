#!/bin/bash
# Counter passes (FETCH_SIZE, WRITE_SIZE, TCC hits / misses) over tools/refetch_ab.py, ONE
# variant per profiled process: the kernels keep their names across variants.
# usage: tools/refetch_pmc.sh <tag> <ops> "<name>:<key=val,...>" ["<name>:..." ...]
# Results: gpurun_out/<tag>_pmc_<name>.txt (per-kernel averages, tools/pmc_summary.py).
set -o pipefail
tag=$1; ops=$2; shift 2
for spec in "$@"; do
  name=${spec%%:*}
  tools/pmc_passes.sh ${tag}_$name traffic python3 tools/refetch_ab.py --variants "$spec" --ops $ops --reps 3 --rounds 1 \
    > gpurun_out/${tag}_pmc_$name.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_$name.log; exit 1; }
  grep -A40 -E "^(kron_pack_kernel<3|rows_ell_kernel<0)" gpurun_out/pmc_${tag}_$name/summary.txt | grep -E "^(kron|rows)|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|_dur_us" > gpurun_out/${tag}_pmc_$name.txt
  cp gpurun_out/pmc_${tag}_$name/summary.txt gpurun_out/${tag}_pmc_${name}_full.txt
  rm -rf gpurun_out/pmc_${tag}_$name
  echo "== $spec"; cat gpurun_out/${tag}_pmc_$name.txt
done
