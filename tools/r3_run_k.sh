#!/bin/bash
# round 3, final evidence on ONE box: flake hunts, full GPU suite, bench lines, per-launch table, rocprof passes, P3.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
export PYTHONUNBUFFERED=1
C=$(cat .commit_stamp 2>/dev/null || echo unknown)
{
echo "# tools/flake_hunt.py on commit $C: the stress test's serving loop (two inputs alternating over four buffer sets,"
echo "# 8 images per batch, XS@256), every collected batch compared with a clean single-stream run"
echo "## final kernel set (no scratch anywhere, dwpw footprint guard): 60000 batches"
timeout 400 python tools/flake_hunt.py --iters 60000 2>&1 | grep -v amdgpu.ids
echo "## round-2 kernel set with the scratch guard lifted (LP_MBT=0 LP_ALLOW_SCRATCH=1 LP_DWPW_GUARD=0): 30000 batches"
LP_MBT=0 LP_ALLOW_SCRATCH=1 LP_DWPW_GUARD=0 timeout 300 python tools/flake_hunt.py --iters 30000 --max-report 6 2>&1 | grep -v amdgpu.ids
echo "## final set without the dwpw footprint guard (LP_DWPW_GUARD=0): 30000 batches"
LP_DWPW_GUARD=0 timeout 300 python tools/flake_hunt.py --iters 30000 --max-report 6 2>&1 | grep -v amdgpu.ids
} > $O/r03_flake_hunt.txt
tail -2 $O/r03_flake_hunt.txt >> $O/summary.txt
timeout 900 python -m pytest tests -v -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; grep -E "FAILED|ERROR|XFAIL|passed|failed" $O/pytest_gpu.log | tail -8 >> $O/summary.txt
timeout 400 python bench.py > $O/r03_bench_n1.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt; grep "timed run\|I/O leg:" $O/bench.err >> $O/summary.txt
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile > $O/r03_bench_n1_200steps.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-S --batch 32 --storage bf16 --no-cpu-baseline > $O/r03_bench_n1_S448_b32_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-S --batch 32 --no-cpu-baseline > $O/r03_bench_n1_S448_b32_f32.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --storage bf16 --no-cpu-baseline > $O/r03_bench_n1_M512_b32_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --arch search-M --size 512 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_n1_M512_b32_f32.json 2>> $O/bench.err
timeout 200 python tools/profile_ops.py --all > $O/r03_per_launch.txt 2>&1
timeout 200 python tools/step_times.py --steps 30 --warmup 5 --stages > $O/r03_step_times.txt 2>&1
timeout 300 python tools/p3_agreement.py --images 64 > $O/r03_p3_agreement.txt 2>&1
{ for mode in thread_local global; do echo "## capture_error_mode=$mode"; timeout 200 python tests/capture_probe.py $mode 2>&1 | grep -v amdgpu.ids | tail -4; done; } > $O/r03_capture_probe.txt
bash tools/evidence.sh r03 $C "per forward of 64 images + 64 mirrored, XS@256, fp32" > $O/evidence.log 2>&1
cp gpurun_out/ev_r03/r03_* $O/ 2>/dev/null
cat $O/summary.txt
