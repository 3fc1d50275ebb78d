#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; tail -3 $O/pytest_gpu.log >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --arch search-S --size 448 --batch 4 --iters 8000 > $O/hunt_S448_f32.txt 2>&1; echo "S448 f32: $(tail -1 $O/hunt_S448_f32.txt)" >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --arch search-S --size 448 --batch 4 --iters 8000 --storage bf16 > $O/hunt_S448_bf16.txt 2>&1; echo "S448 bf16: $(tail -1 $O/hunt_S448_bf16.txt)" >> $O/summary.txt
timeout 300 python tools/flake_hunt.py --arch search-M --size 256 --batch 8 --iters 8000 > $O/hunt_M256_f32.txt 2>&1; echo "M256 f32: $(tail -1 $O/hunt_M256_f32.txt)" >> $O/summary.txt
cat $O/summary.txt; grep -h MISMATCH $O/*.txt | cut -c1-400 | head -10
