#!/bin/bash
# round 3, run Z3: bf16 S@448 b32, switches that exist: tile -> XCD mapping, AE split point, schedule
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z3; mkdir -p $O
export PYTHONUNBUFFERED=1
run() { env "$@" timeout 100 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline --no-io-leg --no-kernel-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['path_roofline']['frac'], d['parity']['ok'])"; }
run LP_NOP=1
run LP_XCD=0
run LP_SPLIT=early
run LP_NOP=2
