#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c10; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu --timeout 600 -x -k "head or stem or every_launch or fused or budget or batched" > $O/pytest_bf16.log 2>&1; tail -12 $O/pytest_bf16.log
for hb in 0 1; do
timeout 300 python tools/profile_ops.py --arch search-S --size 448 --batch 32 --storage bf16 --opt headb=$hb --all | grep -v amdgpu | tail -22 > $O/S448_headb$hb.txt 2>&1
timeout 300 python tools/profile_ops.py --arch search-M --size 512 --batch 32 --storage bf16 --opt headb=$hb --all | grep -v amdgpu | tail -22 > $O/M512_headb$hb.txt 2>&1
done
tail -22 $O/S448_headb0.txt; tail -20 $O/S448_headb1.txt; tail -18 $O/M512_headb1.txt
