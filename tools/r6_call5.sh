#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c6; mkdir -p $O
export LP_NATIVE_FLAVOUR=trace
for o in "mbtb=2 --opt mbtq=0" "mbtb=1 --opt mbtq=0" "mbtb=1 --opt mbtq=2"; do
timeout 200 python tools/wg_timeline.py --arch search-S --size 448 --batch 32 --cexp 96 --opt $o 2>&1 | grep -v amdgpu.ids >> $O/wg_timeline.txt
timeout 200 python tools/wg_timeline.py --arch search-S --size 448 --batch 32 --cexp 192 --opt $o 2>&1 | grep -v amdgpu.ids >> $O/wg_timeline.txt
done
timeout 200 python tools/wg_timeline.py --arch search-S --size 448 --batch 32 --cexp 288 --opt mbtb=2 2>&1 | grep -v amdgpu.ids >> $O/wg_timeline.txt
timeout 200 python tools/wg_timeline.py --arch search-S --size 448 --batch 32 --cexp 720 --opt mbtb=2 2>&1 | grep -v amdgpu.ids >> $O/wg_timeline.txt
cat $O/wg_timeline.txt
