#!/bin/bash
# Round-6 evidence on ONE box: full GPU suite, rocprofv3 passes for the headline and the two bf16 configs, bench lines, per-launch
# tables, stage timings, P3, hunts, the workgroup timeline of the fused bf16 blocks.  Usage: bash tools/final_r06.sh <tag> <commit>
TAG=$1; C=$2
cd $GRAFT_REPO_ROOT
F=gpurun_out/final_$TAG
mkdir -p $F
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 > $F/${TAG}_pytest_gpu.log 2>&1; tail -3 $F/${TAG}_pytest_gpu.log
timeout 900 python bench.py > $F/${TAG}_bench_n1.json 2> $F/bench.err      # the driver's command: headline + BASELINE configs 4 / 5 attached
bash tools/evidence.sh $TAG $C "per forward of 64 images + 64 mirrored, XS@256, fp32" > $F/evidence.log 2>&1
cp gpurun_out/ev_$TAG/${TAG}_* $F/ 2>/dev/null
bash tools/evidence.sh ${TAG}_bf16_S448 $C "per forward of 32 images + 32 mirrored, S@448, bf16 storage" --arch search-S --batch 32 --storage bf16 > $F/evidence_S448.log 2>&1
cp gpurun_out/ev_${TAG}_bf16_S448/${TAG}_bf16_S448_* $F/ 2>/dev/null
bash tools/evidence.sh ${TAG}_bf16_M512 $C "per forward of 32 images + 32 mirrored, M@512, bf16 storage" --arch search-M --size 512 --batch 32 --storage bf16 > $F/evidence_M512.log 2>&1
cp gpurun_out/ev_${TAG}_bf16_M512/${TAG}_bf16_M512_* $F/ 2>/dev/null
timeout 400 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extra-configs > $F/${TAG}_bench_n1_200steps.json 2>> $F/bench.err
timeout 200 python tools/profile_ops.py --all > $F/${TAG}_per_launch.txt 2>&1
timeout 200 python tools/profile_ops.py --all --arch search-S --size 448 --batch 32 --storage bf16 > $F/${TAG}_per_launch_S448_bf16.txt 2>&1
timeout 200 python tools/profile_ops.py --all --arch search-M --size 512 --batch 32 --storage bf16 > $F/${TAG}_per_launch_M512_bf16.txt 2>&1
timeout 200 python tools/step_times.py --steps 30 --warmup 5 --stages > $F/${TAG}_step_times.txt 2>&1
for o in 1 0; do echo "== mbtd=$o" >> $F/${TAG}_mbtd_ab.txt; timeout 200 python tools/profile_ops.py --all --arch search-S --size 448 --batch 32 --storage bf16 --opt mbtd=$o 2>&1 | grep -E "stage.0.1|stage.1.1|^mbt|^total" >> $F/${TAG}_mbtd_ab.txt; timeout 200 python tools/profile_ops.py --all --arch search-M --size 512 --batch 32 --storage bf16 --opt mbtd=$o 2>&1 | grep -E "stage.0.1|stage.1.1|^mbt|^total" >> $F/${TAG}_mbtd_ab.txt; done
timeout 300 python tools/time_plateau.py 2>&1 | grep lp_parse > $F/${TAG}_plateau.txt
timeout 400 python bench.py --config 4 --no-cpu-baseline --parity-images 0 > $F/${TAG}_bench_n1_S448_b32_bf16.json 2>> $F/bench.err
timeout 400 python bench.py --arch search-S --batch 32 --no-cpu-baseline > $F/${TAG}_bench_n1_S448_b32_f32.json 2>> $F/bench.err
timeout 500 python bench.py --config 5 --no-cpu-baseline --parity-images 0 > $F/${TAG}_bench_n1_M512_b32_bf16.json 2>> $F/bench.err
timeout 300 python tools/p3_agreement.py --images 64 > $F/${TAG}_p3_agreement.txt 2>&1
timeout 300 python tools/p3_agreement.py --images 32 --arch search-S --storage bf16 > $F/${TAG}_p3_agreement_bf16.txt 2>&1
H="python tools/flake_hunt.py --max-report 20"
timeout 200 $H --iters 40000 > $F/${TAG}_flake_hunt_XS256_f32_graph.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_XS256_f32_graph.txt | cut -c1-80
timeout 200 $H --iters 40000 --eager > $F/${TAG}_flake_hunt_XS256_f32_eager.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_XS256_f32_eager.txt | cut -c1-80
timeout 200 $H --iters 40000 --arch search-S --size 448 --storage bf16 > $F/${TAG}_flake_hunt_S448_bf16.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_S448_bf16.txt | cut -c1-80
timeout 260 $H --iters 40000 --arch search-M --size 512 --storage bf16 > $F/${TAG}_flake_hunt_M512_bf16.txt 2>&1; tail -1 $F/${TAG}_flake_hunt_M512_bf16.txt | cut -c1-80
export LP_NATIVE_FLAVOUR=trace
W="timeout 200 python tools/wg_timeline.py"
for ce in 96 192 288 720; do $W --arch search-S --size 448 --batch 32 --cexp $ce 2>&1 | grep -v amdgpu.ids >> $F/${TAG}_wg_timeline_S448.txt; done
$W --arch search-M --size 512 --batch 32 --cexp 144 2>&1 | grep -v amdgpu.ids >> $F/${TAG}_wg_timeline_M512.txt
# the three forms of the small residual blocks side by side: mbtd (default), mbtq (mbtd=0), mbtb (mbtd=0, mbtq=0)
for ce in 96 192; do for o in "" "--opt mbtd=0 --opt mbtq=2" "--opt mbtd=0 --opt mbtq=0"; do
  $W --arch search-S --size 448 --batch 32 --cexp $ce $o 2>&1 | grep -v amdgpu.ids >> $F/${TAG}_wg_timeline_mbtd_mbtq_mbtb.txt; done; done
for o in "" "--opt mbtd=0 --opt mbtq=2" "--opt mbtd=0 --opt mbtq=0"; do
  $W --arch search-M --size 512 --batch 32 --cexp 144 $o 2>&1 | grep -v amdgpu.ids >> $F/${TAG}_wg_timeline_mbtd_mbtq_mbtb.txt; done
unset LP_NATIVE_FLAVOUR
for f in bench_n1 bench_n1_200steps bench_n1_S448_b32_bf16 bench_n1_S448_b32_f32 bench_n1_M512_b32_bf16; do
python - $F/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d.get('ms_per_step_200'), d['value'], d['path_roofline']['frac'], d['path_roofline'].get('frac_flops'), d['roofline']['kernel'], d['roofline']['frac'], d['parity']['ok'], d['parity']['p3_vs_pure_cpu_pipeline'].get('oks_vs_cpu_persons'), d.get('latency_ms_single_batch'), d.get('latency_ms_batch1'), d.get('latency_ms_batch8'), {k: (v.get('ms_per_step'), v.get('error')) for k, v in d.get('configs', {}).items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
ls $F
