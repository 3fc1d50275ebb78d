#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
LP_STEM_OCC=1 timeout 100 python tools/profile_ops.py --opt stem=1 --reps 3 2>&1 | grep -E "stem4"
