#!/bin/bash
# round 3, GPU call F: hunt the stress-test failure: 8 runs per kernel configuration, fresh process each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
export PYTHONUNBUFFERED=1
for cfg in "default" "LP_MBT_S2=0" "LP_MBT=0"; do
  for i in 1 2 3 4 5 6 7 8; do
    v=""; [ "$cfg" != "default" ] && v="$cfg"
    env $v timeout 200 python -m pytest tests/test_gpu_real_shapes.py -q -x -k "stress or replay_equals" > $O/t_${cfg//=/_}_$i.log 2>&1
    echo "$cfg $i rc $? $(grep -E 'AssertionError: mismatching|passed|failed' $O/t_${cfg//=/_}_$i.log | tail -2 | tr '\n' ' ')" >> $O/summary.txt
  done
done
cat $O/summary.txt
