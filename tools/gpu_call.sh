#!/bin/bash
# One parametrised GPU call (run from the repository root on the GPU box):
#   tools/gpu_call.sh <tag> <step> [<step> ...]
# Steps run in order and stop at the first failure (never a GPU step after a failed
# or timed-out one).  Results go to gpurun_out/<tag>_*.log.
#   tests[=K]                 pytest -m gpu, optionally -k "K"  (use + for spaces)
#   op=JT,JS[,arith[,tune[,only[,iters[,env]]]]]   tools/op_times.py (tune: k=v;k=v   only: S;P   env: K=V;K=V)
#   launches=JT,JS[,only]     kernel launches per apply: rocprofv3 --stats of op_times with 10
#                             and with 30 iterations, differenced (tools/stats_diff.py)
#   bench[=args]              bench.py (args with + for spaces)
#   round                     tools/profile_round.sh <tag>
#   py=script[,args]          python <script> args (+ for spaces); pyn=name,script[,args]: log named <tag>_<name>.log
#   sh=command                any command (+ for spaces)
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for step in "$@"; do
  name=${step%%=*}; arg=""; [[ "$step" == *=* ]] && arg=${step#*=}
  arg=${arg//+/ }
  case $name in
    tests)
      log=gpurun_out/${tag}_pytest.log
      if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -q -x -k "$arg" > $log 2>&1
      else timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $log 2>&1; fi
      rc=$?; echo "tests rc=$rc"; tail -6 $log; [ $rc -eq 0 ] || exit 1 ;;
    op)
      IFS=, read jt js arith tune only iters envs <<< "$arg"
      log=gpurun_out/${tag}_op_J${jt}_J${js}_${arith:-accurate}${tune:+_${tune//[=;]/_}}${envs:+_${envs//[=;]/_}}.log
      env ${envs//;/ } timeout -k 10 600 python tools/op_times.py --J_time $jt --J_space $js --iters ${iters:-10} \
        --arithmetic ${arith:-accurate} ${tune:+--tune ${tune//;/,}} ${only:+--only ${only//;/,}} 2>&1 | grep -v amdgpu > $log || exit 1
      echo "op J_time=$jt J_space=$js ${arith:-accurate} ${tune}: $(grep -E '^(W|WT|S|P|Kinv|A_x|A_x_packed) ' $log | tr -s ' ' | tr '\n' ';')" ;;
    launches)
      IFS=, read jt js only <<< "$arg"
      for n in 10 30; do
        rm -rf gpurun_out/${tag}_prof$n
        timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof$n -- \
          python3 tools/op_times.py --J_time $jt --J_space $js --iters $n ${only:+--only ${only//;/,}} > gpurun_out/${tag}_prof$n.log 2>&1 || { tail -5 gpurun_out/${tag}_prof$n.log; exit 1; }
        cp $(ls gpurun_out/${tag}_prof$n/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats_$n.csv
        cp $(ls gpurun_out/${tag}_prof$n/*/*kernel_trace.csv | head -1) /tmp/${tag}_kernel_trace_$n.csv
        rm -rf gpurun_out/${tag}_prof$n
      done
      python3 tools/stats_diff.py gpurun_out/${tag}_kernel_stats_10.csv gpurun_out/${tag}_kernel_stats_30.csv 20 \
        > gpurun_out/${tag}_launches_J${jt}_J${js}.txt || exit 1
      python3 tools/trace_diff.py /tmp/${tag}_kernel_trace_10.csv /tmp/${tag}_kernel_trace_30.csv 20 \
        > gpurun_out/${tag}_launches_by_grid_J${jt}_J${js}.txt || exit 1
      head -30 gpurun_out/${tag}_launches_J${jt}_J${js}.txt ;;
    bench)
      timeout -k 10 900 python bench.py $arg > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
      head -c 1500 gpurun_out/${tag}_bench.json; echo ;;
    round)
      tools/profile_round.sh $tag || exit 1 ;;
    py)
      IFS=, read script pargs <<< "$arg"
      log=gpurun_out/${tag}_$(basename ${script%.py}).log
      timeout -k 10 900 python $script $pargs > $log 2>&1 || { tail -15 $log; exit 1; }
      tail -25 $log ;;
    pyn)
      IFS=, read lname script pargs <<< "$arg"
      log=gpurun_out/${tag}_${lname}.log
      timeout -k 10 900 python $script $pargs > $log 2>&1 || { tail -15 $log; exit 1; }
      tail -25 $log ;;
    sh)
      timeout -k 10 900 bash -c "$arg" || exit 1 ;;
    *) echo "unknown step $name"; exit 2 ;;
  esac
done
