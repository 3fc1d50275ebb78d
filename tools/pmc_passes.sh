cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d gpurun_out/pmc1 -o p1 -- python tools/profile_ops.py --reps 1 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc2 -o p2 -- python tools/profile_ops.py --reps 1 > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc3 -o p3 -- python tools/profile_ops.py --reps 1 > gpurun_out/pmc3.log 2>&1
ls gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
