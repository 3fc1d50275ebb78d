#!/bin/bash
# DESIGN 5b diagnostics on the GPU box: does round 3's zero dword come back when dwpw_kernel<3> fetches its bias the old way
# (option diag_dwpw, self-checking), and does it need the LDS-DMA instructions of mbt_kernel (option mbt_dma)?
#   gpurun --timeout 1500 -- bash tools/diag_hunt.sh [iters]
IT=${1:-40000}
O=gpurun_out/diag; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_real_shapes.py -q -x -k "diagnostic_variants" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 400 python tools/flake_hunt.py --iters $IT --diag > $O/h0_dma.txt 2>&1; echo "h0 rc $?"; tail -4 $O/h0_dma.txt
timeout 400 python tools/flake_hunt.py --iters $IT --diag --opt mbt_dma=0 > $O/h1_nodma.txt 2>&1; echo "h1 rc $?"; tail -4 $O/h1_nodma.txt
timeout 400 python tools/flake_hunt.py --iters $IT --diag --eager > $O/h2_dma_eager.txt 2>&1; echo "h2 rc $?"; tail -4 $O/h2_dma_eager.txt
timeout 400 python tools/flake_hunt.py --iters $IT --diag --eager --opt mbt_dma=0 > $O/h3_nodma_eager.txt 2>&1; echo "h3 rc $?"; tail -4 $O/h3_nodma_eager.txt
