#!/bin/bash
# The evidence set of tools/final_round.sh without the long tail (other shapes' full lines, P3 tools, four of the six hunts):
# what must be re-measured after a late kernel change.  Usage: bash tools/final_light.sh <tag> <commit>
TAG=$1; C=$2
cd $GRAFT_REPO_ROOT
F=gpurun_out/final_$TAG
mkdir -p $F
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 > $F/${TAG}_pytest_gpu.log 2>&1; tail -3 $F/${TAG}_pytest_gpu.log
bash tools/evidence.sh $TAG $C "per forward of 64 images + 64 mirrored, XS@256, fp32" > $F/evidence.log 2>&1
cp gpurun_out/ev_$TAG/${TAG}_* $F/ 2>/dev/null
timeout 600 python bench.py > $F/${TAG}_bench_n1.json 2> $F/bench.err
timeout 400 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extra-configs > $F/${TAG}_bench_n1_200steps.json 2>> $F/bench.err
timeout 200 python tools/profile_ops.py --all > $F/${TAG}_per_launch.txt 2>&1
timeout 200 python tools/step_times.py --steps 30 --warmup 5 --stages > $F/${TAG}_step_times.txt 2>&1
H="python tools/flake_hunt.py --max-report 20"
timeout 200 $H --iters 40000 > $F/${TAG}_flake_hunt_XS256_f32_graph.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_XS256_f32_graph.txt | cut -c1-80
timeout 200 $H --iters 40000 --eager > $F/${TAG}_flake_hunt_XS256_f32_eager.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_XS256_f32_eager.txt | cut -c1-80
timeout 200 $H --iters 40000 --arch search-S --size 448 --storage bf16 > $F/${TAG}_flake_hunt_S448_bf16.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_S448_bf16.txt | cut -c1-80
for f in bench_n1 bench_n1_200steps; do
python - $F/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['path_roofline']['frac'], d['path_roofline'].get('frac_flops'), d['roofline']['kernel'], d['roofline']['frac'], d['parity']['ok'], {k: (v.get('ms_per_step'), v.get('error')) for k, v in d.get('configs', {}).items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
ls $F
