#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
export PYTHONUNBUFFERED=1
LP_DWPW_GUARD=1 timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_guard1.txt 2>&1; echo "dwpw footprint 160 regs (no SIMD sharing with mbt): $(tail -1 $O/hunt_guard1.txt)" >> $O/summary.txt
LP_DWPW_GUARD=2 timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_guard2.txt 2>&1; echo "dwpw bias fetched in the epilogue: $(tail -1 $O/hunt_guard2.txt)" >> $O/summary.txt
cat $O/summary.txt; grep -h MISMATCH $O/*.txt | sed -E "s/.*'first_bad_tap': //" | sort | uniq -c | sort -rn | head -20
