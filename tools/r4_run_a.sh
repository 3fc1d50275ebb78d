#!/bin/bash
# round 4, GPU call A: the two micro-benchmarks that decide this round's kernel work
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 120 tools/ubench/bin/phase_mix > $O/phase_mix.txt 2>&1; echo "phase_mix rc $?"
timeout 200 tools/ubench/bin/ldsdma_vs_broadcast 3 > $O/ldsdma_vs_broadcast.txt 2>&1; echo "ldsdma rc $?"
cat $O/phase_mix.txt; cat $O/ldsdma_vs_broadcast.txt
