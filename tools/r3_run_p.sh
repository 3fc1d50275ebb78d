#!/bin/bash
# round 3, run P: pinned LDS requests (dw7.h) in mb16 / mbt / mbt_s2 / mbtb / mbtb_s2; mbt_s2<2,2> back on the path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_real_shapes.py tests/test_gpu_bf16.py -q --timeout 600 -k "mb16_fused or mbt_tiled or xs256_batch64 or native_resolutions or block_taps or fused_bf16 or outputs_vs_fp32" > $O/pytest_sel.log 2>&1; echo "pytest rc $?" >> $O/summary.txt
grep -E "FAILED|ERROR|passed|failed" $O/pytest_sel.log | tail -12 >> $O/summary.txt
timeout 400 python bench.py --no-cpu-baseline > $O/bench_XS256_f32.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_XS256_f32_200.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-S --batch 32 --no-cpu-baseline --no-io-leg > $O/bench_S448_f32.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline --no-io-leg > $O/bench_S448_bf16.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3p/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'), 'io', d.get('io',{}).get('ms_per_step_with_io'))
        for k,v in list(d.get('kernels',{}).items())[:8]: print('    ',k,v['ms_per_step'],v['launches'])
    except Exception as e: print(f, 'ERR', e)
P
cat $O/summary.txt
