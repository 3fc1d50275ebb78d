#!/bin/bash
# Round-2 evidence: traces + PMC passes whose summaries are committed under profiles/.  One counter set per
# pass, every pass under its own timeout.  Usage on the GPU box: bash tools/evidence_r02.sh <commit>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=${1:-unknown}
O=gpurun_out/ev
mkdir -p $O
# (1) single-stream kernel trace of the whole path (network + AE), 3 profiled batches
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace1 -o t -- python tools/run_engine.py --reps 3 --warmup 1 > $O/trace1.log 2>&1
f=$(find $O/trace1 -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/r02_single_stream_kernel_stats.txt
# (2) the bench command itself, traced
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace2 -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-parity-check > $O/trace2.log 2>&1
f=$(find $O/trace2 -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/r02_bench_kernel_stats.txt
# (3) HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes over the same single-stream run
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcf -o p -- python tools/run_engine.py --reps 1 --warmup 1 > $O/pmcf.log 2>&1
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcw -o p -- python tools/run_engine.py --reps 1 --warmup 1 > $O/pmcw.log 2>&1
ff=$(find $O/pmcf -name "*_results.db" | head -1); fw=$(find $O/pmcw -name "*_results.db" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $ff $fw 2 $O/r02_traffic.json "$C" > $O/traffic.log 2>&1
# (4) SQ counters
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmcs -o p -- python tools/run_engine.py --reps 1 --warmup 1 > $O/pmcs.log 2>&1
f=$(find $O/pmcs -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > $O/r02_pmc_sq.txt 2>&1
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmcm -o p -- python tools/run_engine.py --reps 1 --warmup 1 > $O/pmcm.log 2>&1
f=$(find $O/pmcm -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > $O/r02_pmc_mfma_lds.txt 2>&1
ls -la $O/*.txt $O/*.json 2>/dev/null
head -40 $O/r02_single_stream_kernel_stats.txt
