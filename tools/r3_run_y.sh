#!/bin/bash
# round 3, run Y: P3 (GPU pipeline vs the full CPU pipeline) on the round's last build, fp32 and bf16 storage
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3y; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 130 python tools/p3_agreement.py --images 64 > $O/r03_p3_agreement.txt 2>&1; echo "p3 f32 rc $?" >> $O/summary.txt; tail -4 $O/r03_p3_agreement.txt >> $O/summary.txt
timeout 80 python tools/p3_agreement.py --images 32 --storage bf16 > $O/r03_p3_agreement_bf16.txt 2>&1; echo "p3 bf16 rc $?" >> $O/summary.txt; tail -4 $O/r03_p3_agreement_bf16.txt >> $O/summary.txt
cat $O/summary.txt
