#!/bin/bash
# round 3, GPU call A: correctness of the two-workgroup mb16 + the new engine graph cache, A/B bench, per-launch
# table, flake hunt of the replay stress tests under both capture modes, the opt-in bf16 experiments.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; tail -5 $O/pytest_gpu.log >> $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/summary.txt
LP_MB16=4 timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile > $O/bench_mb16_onewg.json 2> $O/bench_mb16_onewg.err
LP_MB16_FENCE=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile > $O/bench_mb16_fence.json 2> $O/bench_mb16_fence.err
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile > $O/bench_200.json 2> $O/bench_200.err
timeout 200 python tools/profile_ops.py --all > $O/per_launch.txt 2>&1
LP_MB16=4 timeout 200 python tools/profile_ops.py --all > $O/per_launch_mb16_onewg.txt 2>&1
timeout 200 python tools/step_times.py --steps 30 --warmup 5 --stages > $O/step_times.txt 2>&1
# flake hunt: the two replay tests + the new graph tests, 6x per capture mode
for mode in global thread_local; do for i in 1 2 3 4 5 6; do
  LP_CAPTURE_MODE=$mode timeout 300 python -m pytest tests/test_gpu_real_shapes.py -q -x -k "replay or stress or survive" > $O/flake_${mode}_$i.log 2>&1
  echo "flake $mode $i rc $? $(tail -1 $O/flake_${mode}_$i.log)" >> $O/summary.txt
done; done
LP_TEST_EXPERIMENTS=1 timeout 600 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16_experiments.log 2>&1; echo "bf16 exp rc $?" >> $O/summary.txt; tail -15 $O/pytest_bf16_experiments.log >> $O/summary.txt
cat $O/summary.txt
