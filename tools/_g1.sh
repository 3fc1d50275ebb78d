cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; mkdir -p gpurun_out/g3
timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/g3/bench.json 2> gpurun_out/g3/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/g3/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('ms_per_step_200'), d.get('latency_ms_single_batch'), d.get('latency_ms_batch1'), d.get('latency_ms_batch8'))
for e in d.get('extra_configs', []): print(e.get('config',{}).get('workload'), e.get('ms_per_step'), e.get('ms_per_step_200'))
PY
