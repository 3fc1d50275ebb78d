#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_default.txt 2>&1; echo "default: $(tail -1 $O/hunt_default.txt)" >> $O/summary.txt
LP_MBT_S2=0 timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_mbt_only.txt 2>&1; echo "mbt only (LP_MBT_S2=0): $(tail -1 $O/hunt_mbt_only.txt)" >> $O/summary.txt
LP_MBT=3 timeout 300 python tools/flake_hunt.py --iters 40000 > $O/hunt_s2_only.txt 2>&1; echo "s2 only (LP_MBT=3): $(tail -1 $O/hunt_s2_only.txt)" >> $O/summary.txt
cat $O/summary.txt; grep -h MISMATCH $O/*.txt | sed -E "s/.*'first_bad_tap': //" | head -40
