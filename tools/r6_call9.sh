#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c9; mkdir -p $O
for rep in 1 2; do for fl in "" prio; do
  if [ -n "$fl" ]; then export LP_NATIVE_FLAVOUR=$fl; else unset LP_NATIVE_FLAVOUR; fi
  echo "== flavour '${fl}' rep $rep" >> $O/young_prio.txt
  timeout 200 python tools/profile_ops.py | tail -10 >> $O/young_prio.txt 2>&1
  timeout 200 python tools/profile_ops.py --arch search-S --size 448 --batch 32 --storage bf16 | tail -9 >> $O/young_prio.txt 2>&1
done; done
unset LP_NATIVE_FLAVOUR
grep -v amdgpu $O/young_prio.txt
