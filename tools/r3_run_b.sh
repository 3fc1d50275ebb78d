#!/bin/bash
# round 3, GPU call B: MFMA || VALU overlap micro-benchmark, the fixed / new tests, bench with the I/O leg.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
export PYTHONUNBUFFERED=1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/mvo tools/ubench/mfma_valu_overlap.hip > $O/mvo_build.log 2>&1
timeout 120 /tmp/mvo > $O/mfma_valu_overlap.txt 2>&1; echo "mvo rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; tail -5 $O/pytest_gpu.log >> $O/summary.txt
timeout 300 python -m pytest tests/test_gpu_real_shapes.py -q -s -k "polling" > $O/polling.log 2>&1; grep "capture under" $O/polling.log >> $O/summary.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/summary.txt
LP_NET_STREAMS=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_1netstream.json 2> $O/bench_1netstream.err
LP_MB16=3 LP_NET_STREAMS=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-io-leg > $O/bench_split_1netstream.json 2> $O/bench_split_1netstream.err
cat $O/summary.txt; cat $O/mfma_valu_overlap.txt
