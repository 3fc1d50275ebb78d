#!/bin/bash
# SQ counter passes (one counter set per pass) over tools/run_engine.py; usage on the GPU box:
#   bash tools/pmc_sq.sh <outdir> [run_engine args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift
mkdir -p $O
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmcs -o p -- python tools/run_engine.py --reps 1 --warmup 1 "$@" > $O/pmcs.log 2>&1
f=$(find $O/pmcs -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > $O/pmc_sq.txt 2>&1
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmcm -o p -- python tools/run_engine.py --reps 1 --warmup 1 "$@" > $O/pmcm.log 2>&1
f=$(find $O/pmcm -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > $O/pmc_mfma_lds.txt 2>&1
rm -rf $O/pmcs $O/pmcm
