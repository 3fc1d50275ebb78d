#!/bin/bash
# round 3, run O: mbtb v3 (pinned LDS requests) + mbtb_s2: fused parity test + bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -s --timeout 600 -k "fused or outputs or engine or batched" > $O/pytest_bf16.log 2>&1; echo "pytest bf16 rc $?" >> $O/summary.txt
grep -E "fused launches|FAILED|ERROR|passed|failed" $O/pytest_bf16.log | tail -24 >> $O/summary.txt
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline --no-io-leg > $O/bench_S448_bf16_mbtb.json 2> $O/bench.err; echo "bench S rc $?" >> $O/summary.txt
timeout 300 python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline --no-io-leg > $O/bench_M512_bf16_mbtb.json 2>> $O/bench.err
timeout 300 python bench.py --storage bf16 --no-cpu-baseline --no-io-leg > $O/bench_XS256_bf16_mbtb.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'))
        for k,v in list(d.get('kernels',{}).items())[:7]: print('    ',k,v['ms_per_step'],v['launches'])
    except Exception as e: print(f, 'ERR', e)
P
cat $O/summary.txt
