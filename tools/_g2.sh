cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
LP_NATIVE_FLAVOUR=gtrace timeout 200 python tools/step_times.py --steps 1 --warmup 0 2>&1 | grep "group n=22" | head -3
