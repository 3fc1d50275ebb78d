#!/bin/bash
# round 3, GPU call D: additive stage merge, fixed I/O leg, full suite; bf16 S@448 variants; r03 evidence passes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|XFAIL|passed|failed" $O/pytest_gpu.log | tail -12 >> $O/summary.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc $?" >> $O/summary.txt; tail -12 $O/bench_default.err >> $O/summary.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_200.json 2> $O/bench_200.err
# bf16 storage, BASELINE config 4 shape: default / dwt 7x7 / dwt 7x7+5x5 / dwtp
for v in "" "LP_DWT=1" "LP_DWT=2" "LP_DWT=1 LP_DWTP=1"; do
  tag=$(echo "$v" | tr ' =' '__'); [ -z "$tag" ] && tag=default
  env $v timeout 400 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline --no-io-leg > $O/bench_S448_bf16_$tag.json 2> $O/bench_S448_bf16_$tag.err
done
timeout 300 python bench.py --arch search-S --batch 32 --no-cpu-baseline --no-io-leg > $O/bench_S448_f32.json 2> $O/bench_S448_f32.err
LP_MBT=0 timeout 300 python bench.py --arch search-S --batch 32 --no-cpu-baseline --no-io-leg --no-kernel-profile > $O/bench_S448_f32_mbt0.json 2> $O/bench_S448_f32_mbt0.err
timeout 200 python tools/profile_ops.py --all > $O/per_launch.txt 2>&1
bash tools/evidence.sh r03 $(cat .commit_stamp 2>/dev/null || echo unknown) "per forward of 64 images + 64 mirrored, XS@256, fp32" > $O/evidence.log 2>&1
cp gpurun_out/ev_r03/r03_* $O/ 2>/dev/null
cat $O/summary.txt
