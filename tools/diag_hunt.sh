#!/bin/bash
# positive control of the hunt on the round's last build: the diagnostic dwpw_kernel<3, ..., DIAG> is the one kernel left with a
# packed fp32 op_sel:[0,1] (its bias add) -- with it on the path (stem = 0) the bias-kind wrong batches must be back
O=gpurun_out/diag10; mkdir -p $O
timeout 120 python tools/flake_hunt.py --max-report 40 --iters 40000 --diag --eager > $O/positive_control_diag_eager.txt 2>&1; grep -c MISMATCH $O/positive_control_diag_eager.txt; tail -2 $O/positive_control_diag_eager.txt | cut -c1-100
timeout 100 python tools/flake_hunt.py --max-report 40 --iters 40000 --storage bf16 --eager > $O/xs_bf16_eager.txt 2>&1; tail -1 $O/xs_bf16_eager.txt | cut -c1-100
