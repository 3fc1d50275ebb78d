#!/bin/bash
# round 3, run M: first hardware run of mbtb_kernel (the fused bf16 InvBottleneck): parity, A/B bench lines, hunt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -s --timeout 600 > $O/pytest_bf16.log 2>&1; echo "pytest bf16 rc $?" >> $O/summary.txt
grep -E "fused launches|FAILED|ERROR|passed|failed" $O/pytest_bf16.log | tail -24 >> $O/summary.txt
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $O/bench_S448_bf16_mbtb.json 2> $O/bench.err; echo "bench S rc $?" >> $O/summary.txt
LP_MBTB=0 timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $O/bench_S448_bf16_chain.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline > $O/bench_M512_bf16_mbtb.json 2>> $O/bench.err
timeout 300 python bench.py --storage bf16 --no-cpu-baseline > $O/bench_XS256_bf16_mbtb.json 2>> $O/bench.err
LP_MBTB=0 timeout 300 python bench.py --storage bf16 --no-cpu-baseline > $O/bench_XS256_bf16_chain.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'))
        for k,v in list(d.get('kernels',{}).items())[:7]: print('    ',k,v['ms_per_step'],v['launches'])
    except Exception as e: print(f, 'ERR', e)
P
timeout 200 python tools/flake_hunt.py --arch search-S --size 448 --batch 4 --storage bf16 --iters 5000 2>&1 | grep -v amdgpu.ids > $O/hunt_S448_bf16.txt; tail -1 $O/hunt_S448_bf16.txt >> $O/summary.txt
cat $O/summary.txt
