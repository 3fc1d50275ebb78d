#!/bin/bash
# round 3, run W: the round's last build: full GPU suite + the committed bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|XFAIL|passed|failed" $O/pytest_gpu.log | tail -8 >> $O/summary.txt
timeout 400 python bench.py > $O/r03_bench_n1.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt; grep "timed run\|I/O leg:" $O/bench.err >> $O/summary.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile > $O/r03_bench_n1_200steps.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-S --batch 32 --no-cpu-baseline --no-io-leg > $O/r03_bench_n1_S448_b32_f32.json 2>> $O/bench.err
timeout 200 python tools/profile_ops.py --all > $O/r03_per_launch.txt 2>&1
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3w/r03_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'), 'io', d.get('io',{}).get('ms_per_step_with_io'), d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'))
    except Exception as e: print(f, 'ERR', e)
P
cat $O/summary.txt
