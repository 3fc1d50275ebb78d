#!/bin/bash
# round 3, run Z2: bf16 S@448 b32 with one / two / four internal network streams (LP_STREAMS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z2; mkdir -p $O
export PYTHONUNBUFFERED=1
for k in 2 1 4; do
LP_STREAMS=$k timeout 100 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline --no-io-leg --no-kernel-profile > $O/b$k.json 2>/dev/null
python - $k <<'P'
import json,sys
d=json.loads(open('gpurun_out/r3z2/b%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); print('LP_STREAMS', sys.argv[1], d['ms_per_step'], d['path_roofline']['frac'], d['parity']['ok'])
P
done
