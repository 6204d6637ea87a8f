#!/bin/bash
# Do the Gauss-Seidel stages pay for partial cache lines?  The same sweeps on 65-step
# slabs (520-byte rows at a 528-byte stride: 5.1 lines per row) and on 64-step slabs
# (512-byte rows: exactly 4 lines), with FETCH_SIZE / WRITE_SIZE / TCC hits per launch.
# usage: tools/gs_line_waste.sh <tag>
set -o pipefail
tag=$1
for n in 65 64; do
  tools/pmc_passes.sh ${tag}_gs$n traffic python3 tools/gs_sweep_time.py 9 $n > gpurun_out/${tag}_gs$n.log 2>&1 || { tail -5 gpurun_out/${tag}_gs$n.log; exit 1; }
  echo "== n_loc=$n"; grep -A9 -E "^rows_ell_kernel<1" gpurun_out/pmc_${tag}_gs$n/summary.txt | grep -E "^rows|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|_dur_us"
  grep -h "per sweep" gpurun_out/pmc_${tag}_gs$n/p1.log | head -4
  cp gpurun_out/pmc_${tag}_gs$n/summary.txt gpurun_out/${tag}_gs${n}_summary.txt
  rm -rf gpurun_out/pmc_${tag}_gs$n
done
