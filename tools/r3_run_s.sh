#!/bin/bash
# round 3, run S: dwpw bias through the scalar cache (GUARD bit 2): eager hunts + repeated stress test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 400 python tools/flake_hunt.py --eager --iters 14000 --max-report 8 2>&1 | grep -v amdgpu.ids > $O/hunt_XS256_f32_eager.txt; tail -4 $O/hunt_XS256_f32_eager.txt | cut -c1-900 >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --eager --storage bf16 --iters 6000 --max-report 8 2>&1 | grep -v amdgpu.ids > $O/hunt_XS256_bf16_eager.txt; tail -3 $O/hunt_XS256_bf16_eager.txt | cut -c1-900 >> $O/summary.txt
for i in 1 2 3 4 5; do timeout 200 python -m pytest tests/test_gpu_real_shapes.py -q --timeout 300 -k "stress or xs256_batch64" 2>&1 | tail -1 >> $O/summary.txt; done
cat $O/summary.txt
