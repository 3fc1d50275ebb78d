#!/bin/bash
# round 4: full GPU suite + bench line with the mb16 runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['path_roofline'], d['roofline'], d['parity'], d.get('latency_ms_single_batch'))
PY
