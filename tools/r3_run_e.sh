#!/bin/bash
# round 3, GPU call E: stride-2 tiled block (mbt_s2_kernel) parity + A/B, I/O leg on the caller's stream, bf16 defaults.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_real_shapes.py -v -k "mbt" --timeout 300 > $O/pytest_mbt.log 2>&1; echo "mbt tests rc $?" >> $O/summary.txt; grep -E "PASS|FAIL|passed|failed|^E  " $O/pytest_mbt.log | head -20 >> $O/summary.txt
for m in 0 1; do LP_MBT_S2=$m timeout 200 python tools/profile_ops.py --all > $O/per_launch_s2_$m.txt 2>&1; done
for m in 0 1; do LP_MBT_S2=$m timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_s2_$m.json 2> $O/bench_s2_$m.err; done
LP_MBT_S2=1 timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_s2_1_200.json 2> $O/bench_s2_1_200.err
timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile > $O/bench_io_main.json 2> $O/bench_io_main.err
LP_IO_OWN_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile > $O/bench_io_own.json 2> $O/bench_io_own.err
LP_IO_OWN_STREAM=1 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile > $O/bench_io_own_q8.json 2> $O/bench_io_own_q8.err
grep "I/O leg\|timed run" $O/bench_io_*.err >> $O/summary.txt
timeout 900 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|XFAIL|passed|failed" $O/pytest_gpu.log | tail -12 >> $O/summary.txt
timeout 400 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline --no-io-leg > $O/bench_S448_bf16.json 2> $O/bench_S448_bf16.err
timeout 400 python bench.py --arch search-S --batch 32 --no-cpu-baseline --no-io-leg > $O/bench_S448_f32.json 2> $O/bench_S448_f32.err
cat $O/summary.txt
