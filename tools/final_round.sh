#!/bin/bash
# End-of-round evidence on ONE box: full GPU test suite, rocprofv3 passes (tools/evidence.sh), bench lines,
# per-launch table, schedule timings, P3.  Usage: bash tools/final_round.sh <tag> <commit>
TAG=$1; C=$2
F=gpurun_out/final_$TAG
mkdir -p $F
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $F/pytest_gpu.txt
bash tools/evidence.sh $TAG $C "per forward of 64 images + 64 mirrored, XS@256, fp32" > $F/evidence.log 2>&1
[ -s gpurun_out/ev_$TAG/${TAG}_traffic.json ] && cp gpurun_out/ev_$TAG/${TAG}_traffic.json profiles/${TAG}_traffic.json
python bench.py > $F/${TAG}_bench_n1.json 2> $F/bench.err
python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $F/${TAG}_bench_n1_200steps.json 2>> $F/bench.err
python tools/profile_ops.py --all > $F/${TAG}_per_launch.txt 2>&1
python tools/step_times.py --steps 30 --warmup 5 --stages > $F/${TAG}_step_times.txt 2>&1
LP_SCHED=lanes python tools/step_times.py --steps 30 --warmup 5 >> $F/${TAG}_step_times.txt 2>&1
python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $F/${TAG}_bench_n1_S448_b32_bf16.json 2>> $F/bench.err
python bench.py --arch search-S --batch 32 --no-cpu-baseline > $F/${TAG}_bench_n1_S448_b32_f32.json 2>> $F/bench.err
python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline > $F/${TAG}_bench_n1_M512_b32_bf16.json 2>> $F/bench.err
python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $F/${TAG}_bench_n1_M512_b32_f32.json 2>> $F/bench.err
python tools/p3_agreement.py --images 64 > $F/${TAG}_p3_agreement.txt 2>&1
ls -la $F
