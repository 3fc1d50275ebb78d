#!/bin/bash
# round 3, run T: final evidence on ONE box: bf16 S@448 rocprof passes, the full GPU suite, the bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
export PYTHONUNBUFFERED=1
C=$(cat .commit_stamp 2>/dev/null || echo unknown)
bash tools/evidence.sh r03_bf16 $C "per forward of 32 images + 32 mirrored, S@448, bf16 storage" --arch search-S --batch 32 --storage bf16 > $O/evidence.log 2>&1
cp gpurun_out/ev_r03_bf16/r03_bf16_* $O/ 2>/dev/null
[ -f $O/r03_bf16_traffic.json ] && cp $O/r03_bf16_traffic.json profiles/r03_traffic_bf16_S448.json
timeout 1200 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|XFAIL|passed|failed" $O/pytest_gpu.log | tail -8 >> $O/summary.txt
timeout 400 python bench.py > $O/r03_bench_n1.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt; grep "timed run\|I/O leg:" $O/bench.err >> $O/summary.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile > $O/r03_bench_n1_200steps.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $O/r03_bench_n1_S448_b32_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline > $O/r03_bench_n1_M512_b32_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --storage bf16 --no-cpu-baseline > $O/r03_bench_n1_XS256_b64_bf16.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3t/r03_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), 'net single stream', d.get('network_ms_single_stream'), 'io', d.get('io',{}).get('ms_per_step_with_io'), 'traffic', d.get('roofline',{}).get('traffic'), d.get('roofline',{}).get('kernel'))
    except Exception as e: print(f, 'ERR', e)
P
cat $O/summary.txt
