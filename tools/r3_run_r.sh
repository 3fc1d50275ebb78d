#!/bin/bash
# round 3, run R: is the residual wrong batch an EAGER-launch event?  (stress test failed at batch 1 of 48 = an eager stage)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 400 python tools/flake_hunt.py --eager --iters 12000 --max-report 8 2>&1 | grep -v amdgpu.ids > $O/hunt_XS256_f32_eager.txt; tail -9 $O/hunt_XS256_f32_eager.txt >> $O/summary.txt
for i in 1 2 3 4 5 6; do timeout 120 python tools/flake_hunt.py --iters 400 --max-report 3 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/hunt_XS256_f32_short_runs.txt; done
cat $O/hunt_XS256_f32_short_runs.txt >> $O/summary.txt
cat $O/summary.txt
