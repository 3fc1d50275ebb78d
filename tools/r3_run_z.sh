#!/bin/bash
# round 3, run Z: completion timestamps of the serving loop and the NET / AE graphs alone on the last build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 100 python tools/step_times.py --steps 30 --warmup 5 --stages > $O/r03_step_times.txt 2>&1
tail -5 $O/r03_step_times.txt
