#!/bin/bash
# Round 4: every fused-block kernel owns its CU (LP_OWN_CU).  Hunts of the product (register staging) and of the dma flavour.
IT=${1:-40000}
O=gpurun_out/diag5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_real_shapes.py -q -x -k "stem or diagnostic or flavour or mb16 or mbt" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 120 tools/ubench/bin/stem4_trace > $O/stem4_trace.txt 2>&1; echo "trace rc $?"; sed -n 5,24p $O/stem4_trace.txt
timeout 200 python tools/profile_ops.py > $O/ops.txt 2>&1; tail -11 $O/ops.txt
LP_NATIVE_FLAVOUR=dma timeout 200 python tools/profile_ops.py > $O/ops_dma.txt 2>&1; tail -11 $O/ops_dma.txt | head -3
H="timeout 400 python tools/flake_hunt.py --iters $IT --max-report 40"
run() { name=$1; shift; "$@" > $O/$name.txt 2>&1; echo "$name rc $? mismatches $(grep -c MISMATCH $O/$name.txt)"; tail -1 $O/$name.txt | cut -c1-120; }
run g1_eager $H --eager
run g2_graph $H
run g3_bf16_graph $H --storage bf16
run g4_diag_eager $H --diag --eager
LP_NATIVE_FLAVOUR=dma run g5_dma_eager $H --eager
LP_NATIVE_FLAVOUR=dma run g6_dma_graph $H
LP_NATIVE_FLAVOUR=dma run g7_dma_bf16_graph $H --storage bf16
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/diag5/bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d['path_roofline']['frac'], d['roofline']['kernel'], d['roofline']['frac'], d['parity']['ok'], d.get('latency_ms_single_batch'))
P
