#!/bin/bash
# round 3, GPU call H: is scratch the cause of the rare wrong batch?  default (no spilling kernel), the old kernel
# set behind the guard (spilling variants refused), the old kernel set with the guard lifted.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_default.txt 2>&1; echo "default: $(tail -1 $O/hunt_default.txt)" >> $O/summary.txt
LP_MBT=0 timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_mbt0_guard.txt 2>&1; echo "LP_MBT=0 guarded: $(tail -1 $O/hunt_mbt0_guard.txt)" >> $O/summary.txt
LP_MBT=0 LP_ALLOW_SCRATCH=1 timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_mbt0_scratch.txt 2>&1; echo "LP_MBT=0 LP_ALLOW_SCRATCH=1: $(tail -1 $O/hunt_mbt0_scratch.txt)" >> $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline --no-io-leg > $O/bench.json 2> $O/bench.err; grep "timed run" $O/bench.err >> $O/summary.txt
timeout 900 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|XFAIL|passed|failed" $O/pytest_gpu.log | tail -12 >> $O/summary.txt
cat $O/summary.txt
