#!/bin/bash
# Set-up timeline (tools/setup_profile.py --timeline) under several interpreter switch
# intervals: do the planner threads lose time handing the interpreter lock around?
#   tools/switch_interval_ab.sh <tag>
tag=$1
cd "$GRAFT_REPO_ROOT" || exit 1
for iv in default 0.0005 0.00005 default; do
  log=gpurun_out/${tag}_switch_${iv}_$RANDOM.log
  if [ $iv = default ]; then python tools/setup_profile.py --timeline > $log 2>&1 || exit 1
  else STK_SWITCH_INTERVAL=$iv python tools/setup_profile.py --timeline > $log 2>&1 || exit 1; fi
  echo "$iv: $(grep '^set-up' $log | tr '\n' ' ')"
done
