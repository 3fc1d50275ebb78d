#!/bin/bash
# round 3, GPU call C: the tiled fused block (mbt_kernel) -- parity, per-launch times, bench A/B; bench with the
# I/O leg and progress log; capture probe; full suite.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_real_shapes.py -v -k "mbt or mb16" --timeout 300 > $O/pytest_mbt.log 2>&1; echo "mbt tests rc $?" >> $O/summary.txt; grep -E "PASS|FAIL|passed|failed" $O/pytest_mbt.log | tail -12 >> $O/summary.txt
for m in 0 1 2; do LP_MBT=$m timeout 200 python tools/profile_ops.py --all > $O/per_launch_mbt$m.txt 2>&1; done
for m in 0 1 2; do LP_MBT=$m timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_mbt$m.json 2> $O/bench_mbt$m.err; done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc $?" >> $O/summary.txt
for mode in thread_local global; do timeout 200 python tests/capture_probe.py $mode > $O/probe_$mode.json 2> $O/probe_$mode.err; echo "probe $mode rc $? $(tail -1 $O/probe_$mode.json)" >> $O/summary.txt; done
timeout 400 python -m pytest tests/test_gpu_real_shapes.py -v -x -s -k "world2" --timeout 380 > $O/pytest_world2.log 2>&1; echo "world2 rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|passed|failed" $O/pytest_gpu.log | tail -12 >> $O/summary.txt
cat $O/summary.txt
