#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c4; mkdir -p $O
for cfg in "--arch search-S --size 448 --batch 12" "--arch search-S --size 208 --batch 40" "--arch search-M --size 512 --batch 6" "--arch search-XS --size 256 --batch 33"; do
  echo "== $cfg" >> $O/check.txt; timeout 300 python tools/block_ab_check.py $cfg 2>&1 | grep -v amdgpu.ids >> $O/check.txt
done
for o in "mbtb=2 --opt mbtq=0" "mbtb=1 --opt mbtq=0" "mbtb=1 --opt mbtq=1"; do
n=$(echo $o | tr -d ' =-' )
timeout 300 python tools/profile_ops.py --arch search-S --size 448 --batch 32 --storage bf16 --all --opt $o > $O/per_launch_S448_$n.txt 2>&1
timeout 300 python tools/profile_ops.py --arch search-M --size 512 --batch 32 --storage bf16 --all --opt $o > $O/per_launch_M512_$n.txt 2>&1
done
export LP_NATIVE_FLAVOUR=trace
timeout 300 python tools/phase_trace.py --arch search-S --size 448 --batch 32 --opt mbtq=0 2>&1 | grep -v amdgpu.ids >> $O/phase_trace.txt
timeout 300 python tools/phase_trace.py --arch search-M --size 512 --batch 32 --opt mbtq=0 2>&1 | grep -v amdgpu.ids >> $O/phase_trace.txt
unset LP_NATIVE_FLAVOUR
cat $O/check.txt | grep -v "elements 0 of"; for f in $O/per_launch_*; do echo $f; grep "stage.0.1\|stage.1.1\|^mbt\|^total" $f; done; cat $O/phase_trace.txt
