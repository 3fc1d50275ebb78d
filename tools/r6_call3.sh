#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c3; mkdir -p $O
export LP_NATIVE_FLAVOUR=trace
for q in 0 2; do
timeout 300 python tools/phase_trace.py --arch search-S --size 448 --batch 32 --opt mbtq=$q >> $O/phase_trace.txt 2>&1
timeout 300 python tools/phase_trace.py --arch search-M --size 512 --batch 32 --opt mbtq=$q >> $O/phase_trace.txt 2>&1
done
unset LP_NATIVE_FLAVOUR
# small-batch latency of the fp32 headline path: per-launch at batch 1 / 8 with and without mb16
for b in 1 8; do for o in "mb16=1" "mb16=0" "mb16=0 --opt mbt=0"; do
echo "== batch $b $o" >> $O/small_batch.txt
timeout 200 python tools/profile_ops.py --batch $b --opt $o | tail -12 >> $O/small_batch.txt 2>&1
done; done
cat $O/phase_trace.txt | grep -v amdgpu.ids; cat $O/small_batch.txt
