#!/bin/bash
# round 3, run V: pinned LDS requests (dw7.h) in the fp32 fused kernels, register footprints >= the hunted build's,
# mbt_s2<2,2> back on the path -- parity, hunts (graph + eager), the long-process scenario, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_real_shapes.py -q --timeout 600 -k "mb16_fused or mbt_tiled or block_taps or native_resolutions or fused_stem" > $O/pytest_parity.log 2>&1; echo "parity rc $?" >> $O/summary.txt; tail -2 $O/pytest_parity.log >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --iters 20000 --max-report 6 2>&1 | grep -v amdgpu.ids > $O/hunt_graph.txt; tail -3 $O/hunt_graph.txt | cut -c1-700 >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --eager --iters 8000 --max-report 6 2>&1 | grep -v amdgpu.ids > $O/hunt_eager.txt; tail -3 $O/hunt_eager.txt | cut -c1-700 >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_real_shapes.py -q --timeout 600 -k "fused_bf16 or xs256_batch64 or stress" > $O/pytest_long.log 2>&1; echo "long-process rc $?" >> $O/summary.txt; tail -2 $O/pytest_long.log >> $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_XS256_f32.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_XS256_f32_200.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3v/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'), 'io', d.get('io',{}).get('ms_per_step_with_io'))
        for k,v in list(d.get('kernels',{}).items())[:6]: print('    ',k,v['ms_per_step'],v['launches'])
    except Exception as e: print(f, 'ERR', e)
P
cat $O/summary.txt
