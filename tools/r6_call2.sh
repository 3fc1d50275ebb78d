#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c2; mkdir -p $O
for cfg in "--arch search-S --size 448 --batch 4" "--arch search-S --size 200 --batch 3" "--arch search-M --size 512 --batch 2" "--arch search-XS --size 256 --batch 5"; do
  echo "== $cfg" >> $O/check.txt; timeout 300 python tools/r6_mbtq_check.py $cfg >> $O/check.txt 2>&1
done
for q in 0 1; do
timeout 300 python tools/profile_ops.py --arch search-S --size 448 --batch 32 --storage bf16 --all --opt mbtq=$q > $O/per_launch_S448_bf16_q$q.txt 2>&1
timeout 300 python tools/profile_ops.py --arch search-M --size 512 --batch 32 --storage bf16 --all --opt mbtq=$q > $O/per_launch_M512_bf16_q$q.txt 2>&1
done
cat $O/check.txt; for f in $O/per_launch_*; do echo $f; grep "stage.0.1\|stage.1.1\|^mbt\|^total" $f; done
