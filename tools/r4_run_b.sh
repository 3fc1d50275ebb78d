#!/bin/bash
# round 4, GPU call B: mb16p_kernel correctness + timing experiments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/mb16p_check.py --archs search-XS --dbg1 1,2,4,8,6,7,15,16,31 > $O/mb16p_check.txt 2>&1; echo "check rc $?"
tail -30 $O/mb16p_check.txt
