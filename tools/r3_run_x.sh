#!/bin/bash
# round 3, run X: rocprof passes of the round's last build (XS@256 b64 fp32) and the bench lines that quote them
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x; mkdir -p $O
export PYTHONUNBUFFERED=1
C=$(cat .commit_stamp 2>/dev/null || echo unknown)
bash tools/evidence.sh r03 $C "per forward of 64 images + 64 mirrored, XS@256, fp32" > $O/evidence.log 2>&1
cp gpurun_out/ev_r03/r03_* $O/ 2>/dev/null
[ -f $O/r03_traffic.json ] && cp $O/r03_traffic.json profiles/r03_traffic.json
timeout 400 python bench.py > $O/r03_bench_n1.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt; grep "timed run\|I/O leg:" $O/bench.err >> $O/summary.txt
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $O/r03_bench_n1_S448_b32_bf16.json 2>> $O/bench.err
python - <<'P' >> $O/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r3x/r03_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], 'ms  path frac', d['path_roofline']['frac'], 'parity', d.get('parity',{}).get('ok'), d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('traffic'), d.get('roofline',{}).get('traffic_source'))
    except Exception as e: print(f, 'ERR', e)
P
grep -n "mb16_kernel" $O/r03_single_stream_kernel_stats.txt | head -4 >> $O/summary.txt
cat $O/summary.txt
