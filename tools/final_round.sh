#!/bin/bash
# End-of-round evidence on ONE box: full GPU test suite, rocprofv3 passes (tools/evidence.sh), bench lines,
# per-launch table, schedule timings, P3.  Usage: bash tools/final_round.sh <tag> <commit>
TAG=$1; C=$2
cd $GRAFT_REPO_ROOT
F=gpurun_out/final_$TAG
mkdir -p $F
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 > $F/${TAG}_pytest_gpu.log 2>&1; tail -3 $F/${TAG}_pytest_gpu.log
bash tools/evidence.sh $TAG $C "per forward of 64 images + 64 mirrored, XS@256, fp32" > $F/evidence.log 2>&1
cp gpurun_out/ev_$TAG/${TAG}_* $F/ 2>/dev/null
timeout 600 python bench.py > $F/${TAG}_bench_n1.json 2> $F/bench.err      # the driver's command: headline + BASELINE configs 4 / 5 attached
LP_AE=dm timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-io-leg > $F/${TAG}_bench_n1_ae_dm_path.json 2>> $F/bench.err   # rounds 2-4 AE path, same box
timeout 400 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extra-configs > $F/${TAG}_bench_n1_200steps.json 2>> $F/bench.err
timeout 200 python tools/profile_ops.py --all > $F/${TAG}_per_launch.txt 2>&1
timeout 200 python tools/step_times.py --steps 30 --warmup 5 --stages > $F/${TAG}_step_times.txt 2>&1
timeout 400 python bench.py --config 4 --no-cpu-baseline > $F/${TAG}_bench_n1_S448_b32_bf16.json 2>> $F/bench.err
timeout 400 python bench.py --arch search-S --batch 32 --no-cpu-baseline > $F/${TAG}_bench_n1_S448_b32_f32.json 2>> $F/bench.err
timeout 500 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > $F/${TAG}_bench_n1_M512_b32_bf16.json 2>> $F/bench.err
timeout 300 python tools/p3_agreement.py --images 64 > $F/${TAG}_p3_agreement.txt 2>&1
timeout 300 python tools/p3_agreement.py --images 32 --arch search-S --storage bf16 > $F/${TAG}_p3_agreement_bf16.txt 2>&1
timeout 100 python tools/time_tta.py > $F/${TAG}_tta_merge_kernels.txt 2>&1
timeout 90 tools/ubench/bin/pk_vs_mfma 1 > $F/${TAG}_pk_vs_mfma.txt 2>&1
tools/ubench/bin/dpp_check > $F/${TAG}_dpp_check.txt 2>&1
# hunts of this build on the larger shapes (XS@256 fp32 eager / graph, XS@256 bf16 and 20 000 of S@448 fp32: tools/diag_hunt.sh)
H="python tools/flake_hunt.py --max-report 20"
timeout 200 $H --iters 40000 > $F/${TAG}_flake_hunt_XS256_f32_graph.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_XS256_f32_graph.txt | cut -c1-80
timeout 200 $H --iters 40000 --eager > $F/${TAG}_flake_hunt_XS256_f32_eager.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_XS256_f32_eager.txt | cut -c1-80
timeout 200 $H --iters 40000 --arch search-S --size 448 --storage bf16 > $F/${TAG}_flake_hunt_S448_bf16.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_S448_bf16.txt | cut -c1-80
timeout 260 $H --iters 40000 --arch search-M --size 512 --storage bf16 > $F/${TAG}_flake_hunt_M512_bf16.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_M512_bf16.txt | cut -c1-80
timeout 160 $H --iters 20000 --arch search-S --size 448 > $F/${TAG}_flake_hunt_S448_f32.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_S448_f32.txt | cut -c1-80
timeout 260 $H --iters 20000 --arch search-M --size 512 > $F/${TAG}_flake_hunt_M512_f32.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_M512_f32.txt | cut -c1-80
for f in bench_n1 bench_n1_200steps bench_n1_S448_b32_bf16 bench_n1_S448_b32_f32 bench_n1_M512_b32_bf16; do
python - $F/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['path_roofline']['frac'], d['path_roofline'].get('frac_flops'), d['roofline']['kernel'], d['roofline']['frac'], d['parity']['ok'], d['parity']['p3_vs_pure_cpu_pipeline'].get('oks_vs_cpu_persons'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
ls $F
