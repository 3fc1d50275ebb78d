#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/flake_hunt.py --iters 30000 > $O/hunt_default.txt 2>&1; tail -3 $O/hunt_default.txt >> $O/summary.txt
LP_MBT=0 timeout 300 python tools/flake_hunt.py --iters 30000 > $O/hunt_mbt0.txt 2>&1; tail -3 $O/hunt_mbt0.txt >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --iters 8000 --eager > $O/hunt_eager.txt 2>&1; tail -3 $O/hunt_eager.txt >> $O/summary.txt
LP_NET_STREAMS=1 timeout 300 python tools/flake_hunt.py --iters 20000 > $O/hunt_1net.txt 2>&1; tail -3 $O/hunt_1net.txt >> $O/summary.txt
cat $O/summary.txt; grep MISMATCH $O/*.txt | head -40
