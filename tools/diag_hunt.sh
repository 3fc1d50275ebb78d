#!/bin/bash
# Round 4: the library without packed-fp32 op_sel:[0,1] forms (LDS-DMA staging, no CU claims): parity + hunts
IT=${1:-40000}
O=gpurun_out/diag8; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_real_shapes.py tests/test_gpu_bf16.py -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
H="timeout 400 python tools/flake_hunt.py --max-report 40"
run() { name=$1; shift; "$@" > $O/$name.txt 2>&1; echo "$name rc $? mismatches $(grep -c MISMATCH $O/$name.txt)"; tail -1 $O/$name.txt | cut -c1-120; }
run n1_eager $H --iters $IT --eager
run n2_graph $H --iters $IT
run n3_bf16_graph $H --iters $IT --storage bf16
run n4_s448_f32_graph $H --iters 20000 --arch search-S --size 448
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/diag8/bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], d['path_roofline']['frac'], d['roofline']['kernel'], d['roofline']['frac'], d['parity']['ok'], d.get('latency_ms_single_batch'))
P
