#!/bin/bash
# round 3, run U: sanity of the last host-side changes (tap of unstored tensors fails loudly; lp_parse_dm refine error)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_valid_loop.py -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/summary.txt; tail -3 $O/pytest.log >> $O/summary.txt
python - <<'P' >> $O/summary.txt 2>&1
import torch
from oracle import synth
from litepose_amd import arch_zoo, config
from litepose_amd.models import pose_mobilenet
arch = arch_zoo.get('search-XS'); sd = synth.make_state_dict(arch, seed=1)
m = pose_mobilenet.get_pose_net(config.get_cfg('crowd_pose'), cfg_arch=arch, storage='bf16'); m.load_state_dict(sd, strict=True)
m.forward_native(synth.make_images(1, 128, seed=2).cuda(), 0)
print('block tap elems', m.tap('stage.1.2').numel())
try:
    m.tap('stage.1.2.inv'); print('inner tap: NO ERROR (wrong)')
except Exception as e:
    print('inner tap raises:', str(e)[:120])
import __graft_entry__ as g
g.smoke(); print('smoke ok')
P
cat $O/summary.txt
