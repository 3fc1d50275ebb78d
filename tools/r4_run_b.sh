#!/bin/bash
# round 4, GPU call: mb16 runs + the extended LDS-DMA reproducer
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/mb16_check.py > $O/mb16_check.txt 2>&1; echo "check rc $?"
tail -30 $O/mb16_check.txt
timeout 200 tools/ubench/bin/ldsdma_vs_broadcast 3 > $O/ldsdma_vs_broadcast.txt 2>&1; echo "ldsdma rc $?"
tail -12 $O/ldsdma_vs_broadcast.txt
