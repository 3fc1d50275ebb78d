#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c7; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 -x > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6c7/bench_n1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('ms_per_step_200'), d['value'], d['latency_ms_single_batch'], d.get('latency_ms_batch1'), d.get('latency_ms_batch8'))
print(d['roofline'].get('cus_occupied'), d['roofline'].get('frac_flops_per_occupied_cu'), d['path_roofline'].get('frac_flops'), d['path_roofline'].get('frac_flops_bf16x3'))
print({k:(v.get('ms_per_step'),v.get('ms_per_step_200'),v.get('wall_s'),v.get('parity',{}).get('images'),v.get('error')) for k,v in d['configs'].items()})
PY
