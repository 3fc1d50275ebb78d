#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c8; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu --timeout 600 -x -k "stem or fused or every_launch or mbtq or budget" > $O/pytest_bf16.log 2>&1; tail -12 $O/pytest_bf16.log
for st in 0 1; do
timeout 300 python tools/profile_ops.py --arch search-S --size 448 --batch 32 --storage bf16 --opt stem=$st | tail -12 > $O/S448_stem$st.txt 2>&1
timeout 300 python tools/profile_ops.py --arch search-M --size 512 --batch 32 --storage bf16 --opt stem=$st | tail -12 > $O/M512_stem$st.txt 2>&1
done
tail -12 $O/S448_stem0.txt $O/S448_stem1.txt $O/M512_stem1.txt
timeout 300 python tools/time_ae.py > $O/time_ae.txt 2>&1; grep -i "parse\|group" $O/time_ae.txt
