#!/bin/bash
# round 6, GPU call 1: dot2 micro-benchmark + same-box baselines (per-launch bf16 S@448 / M@512, default bench line)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c1; mkdir -p $O
./tools/ubench/bin/dot2_rate > $O/dot2_rate.txt 2>&1
timeout 300 python tools/profile_ops.py --arch search-S --size 448 --batch 32 --storage bf16 --all > $O/per_launch_S448_bf16.txt 2>&1
timeout 300 python tools/profile_ops.py --arch search-M --size 512 --batch 32 --storage bf16 --all > $O/per_launch_M512_bf16.txt 2>&1
timeout 300 python tools/profile_ops.py --all > $O/per_launch_XS256_f32.txt 2>&1
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench.err
cat $O/dot2_rate.txt; tail -12 $O/per_launch_S448_bf16.txt; tail -c 600 $O/bench_n1.json
