#!/bin/bash
# PMC passes over one profiled forward (tools/profile_ops.py --reps 1), one counter set per pass,
# each wrapped in its own timeout (a failed counter config must not hang the box).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*(MFMA|LDS|BARRIER|VALU|WAIT|BUSY_CY)[A-Z_0-9]*" | sort -u > gpurun_out/pmc_names.txt
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"
i=1
for P in "$P1" "$P2"; do
  timeout 240 rocprofv3 --kernel-trace --pmc $P -d gpurun_out/pmcx$i -o p -- python tools/profile_ops.py --reps 1 > gpurun_out/pmcx$i.log 2>&1
  echo "pass $i rc=$?"
  f=$(find gpurun_out/pmcx$i -name "*_results.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_pmc.py $f > gpurun_out/pmcx$i.txt 2>&1
  i=$((i+1))
done
grep -E "kernel|mb16|mbconv" gpurun_out/pmcx1.txt | cut -c1-250
grep -E "kernel|mb16|mbconv" gpurun_out/pmcx2.txt | cut -c1-250
