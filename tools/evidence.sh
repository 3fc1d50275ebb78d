#!/bin/bash
# Evidence whose summaries are committed under profiles/: kernel traces + PMC passes, one counter set per pass,
# every pass under its own timeout.  Usage on the GPU box:
#   bash tools/evidence.sh <tag> <commit> <note> [run_engine args...]      e.g.  r02 abc1234 "XS@256 b64 f32"
#   bash tools/evidence.sh r02_bf16 abc1234 "S@448 b32 bf16" --arch search-S --batch 32 --storage bf16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; C=$2; NOTE=$3; shift 3
O=gpurun_out/ev_$TAG
mkdir -p $O
# the configuration the passes run on (run_engine.py's defaults: XS@256 b64 f32) -> stamped into the traffic file
CFG=$(python - "$@" <<'PY'
import json, sys
a = sys.argv[1:]
g = lambda k, d: a[a.index(k) + 1] if k in a else d
from litepose_amd import arch_zoo
arch = g('--arch', 'search-XS')
print(json.dumps({'arch': arch, 'size': int(g('--size', arch_zoo.get(arch)['img_size'])), 'batch': int(g('--batch', 64)),
                  'storage': g('--storage', 'f32')}))
PY
)
# (1) single-stream kernel trace of the whole path (network + AE), 3 profiled batches
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace1 -o t -- python tools/run_engine.py --reps 3 --warmup 1 "$@" > $O/trace1.log 2>&1
f=$(find $O/trace1 -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/${TAG}_single_stream_kernel_stats.txt
# (2) HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes over the same single-stream run
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcf -o p -- python tools/run_engine.py --reps 1 --warmup 1 "$@" > $O/pmcf.log 2>&1
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcw -o p -- python tools/run_engine.py --reps 1 --warmup 1 "$@" > $O/pmcw.log 2>&1
ff=$(find $O/pmcf -name "*_results.db" | head -1); fw=$(find $O/pmcw -name "*_results.db" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $ff $fw 2 $O/${TAG}_traffic.json "$C" "$NOTE" "$CFG" > $O/traffic.log 2>&1
# (3) SQ counters
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmcs -o p -- python tools/run_engine.py --reps 1 --warmup 1 "$@" > $O/pmcs.log 2>&1
f=$(find $O/pmcs -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > $O/${TAG}_pmc_sq.txt 2>&1
LP_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmcm -o p -- python tools/run_engine.py --reps 1 --warmup 1 "$@" > $O/pmcm.log 2>&1
f=$(find $O/pmcm -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py $f > $O/${TAG}_pmc_mfma_lds.txt 2>&1
rm -rf $O/trace1 $O/pmcf $O/pmcw $O/pmcs $O/pmcm
ls -la $O
