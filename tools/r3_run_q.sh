#!/bin/bash
# round 3, run Q: fp32 kernels back at the hunted set; mbtb / mbtb_s2 with pinned LDS requests + whole-register-budget guard
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_real_shapes.py -q -s --timeout 600 -k "bf16 or xs256_batch64 or stress or no_wrong_batch" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/summary.txt
grep -E "fused launches|FAILED|ERROR|passed|failed" $O/pytest_sel.log | tail -24 >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --storage bf16 --iters 10000 2>&1 | grep -v amdgpu.ids > $O/hunt_XS256_bf16.txt; tail -1 $O/hunt_XS256_bf16.txt >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --arch search-S --size 448 --batch 4 --storage bf16 --iters 6000 2>&1 | grep -v amdgpu.ids > $O/hunt_S448_bf16.txt; tail -1 $O/hunt_S448_bf16.txt >> $O/summary.txt
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $O/bench_S448_bf16.json 2> $O/bench.err; echo "bench S rc $?" >> $O/summary.txt
timeout 300 python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline > $O/bench_M512_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --storage bf16 --no-cpu-baseline > $O/bench_XS256_bf16.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3q/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'), 'io', d.get('io',{}).get('ms_per_step_with_io'))
        for k,v in list(d.get('kernels',{}).items())[:8]: print('    ',k,v['ms_per_step'],v['launches'])
    except Exception as e: print(f, 'ERR', e)
P
cat $O/summary.txt
