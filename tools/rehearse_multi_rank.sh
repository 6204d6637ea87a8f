set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; n=$2; shift 2; STK_BACKEND=gloo OMP_NUM_THREADS=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) "$@" > gpurun_out/r06_rehearse_$name.log 2>&1; rc=$?; echo "$name rc=$rc: $(grep -E 'Completed|Total|Final r.Pr' gpurun_out/r06_rehearse_$name.log | tr '\n' ' ' | cut -c1-220)"; [ $rc -eq 0 ] || { tail -5 gpurun_out/r06_rehearse_$name.log; exit 1; }; }
run timing_original_n4 4 spacetime-fullgrid-parallel_amd/heateq_mpi_timing.py --J_time 6 --J_space 8 --iters 3
run timing_composite_n3 3 spacetime-fullgrid-parallel_amd/heateq_mpi_timing.py --J_time 5 --J_space 8 --iters 3 --wavelettransform composite
run solve_interleaved_n4 4 spacetime-fullgrid-parallel_amd/heateq_mpi.py --J_time 5 --J_space 8 --wavelettransform interleaved
run solve_lshape_n5 5 spacetime-fullgrid-parallel_amd/heateq_mpi.py --J_time 5 --J_space 8 --problem lshape
run solve_reference_schur_n2 2 spacetime-fullgrid-parallel_amd/heateq_mpi.py --J_time 5 --J_space 7 --schur reference
run solve_reference_arithmetic_n3 3 spacetime-fullgrid-parallel_amd/heateq_mpi.py --J_time 5 --J_space 7 --arithmetic reference
