#!/bin/bash
# Everything the round's numbers come from, in one GPU call (run from the repo
# root on the GPU box; results under gpurun_out/<tag>/, copy what is to be judged
# into profiles/):
#   1. bench.py as the driver runs it (default flags)            -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command        -> kernel_stats.csv
#   3. PMC passes on the Kronecker kernel + the traffic record     -> pmc_kron.txt, pmc_traffic.json
#   4. PMC passes on the solve (Gauss-Seidel kernels)              -> pmc_solve.txt
# usage: tools/profile_round.sh <tag>
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
echo "bench done"; cat $out/bench.json | head -c 600; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/prof.log || { tail -5 $out/prof.log; exit 1; }
cp $(ls $out/prof/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 tools/trace_gaps.py $(ls $out/prof/*/*kernel_trace.csv | head -1) kron_pack_kernel > $out/kron_launch_gaps.txt 2>&1 || true
rm -rf $out/prof   # the raw trace is large; gpurun_out is capped at 64 MiB
echo "kernel stats done"; head -4 $out/kernel_stats.csv; cat $out/kron_launch_gaps.txt
tools/pmc_passes.sh ${tag}_kron kron python3 tools/kron_one.py --kernels packed > $out/pmc_kron.log 2>&1
cp gpurun_out/pmc_${tag}_kron/summary.txt $out/pmc_kron.txt
python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_kron kron_pack_kernel $out/pmc_traffic.json > $out/pmc_traffic.log 2>&1
rm -rf gpurun_out/pmc_${tag}_kron
echo "kron pmc done"; cat $out/pmc_traffic.json | head -c 400; echo
tools/pmc_passes.sh ${tag}_solve gs python3 bench.py --steps 2 --warmup 1 --solve-iters 2 --no-cpu-baseline --preheat 0 > $out/pmc_solve.log 2>&1
cp gpurun_out/pmc_${tag}_solve/summary.txt $out/pmc_solve.txt
rm -rf gpurun_out/pmc_${tag}_solve
echo "solve pmc done"
