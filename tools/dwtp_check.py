"""One-shot hardware check of dwtp_kernel (LP_DWTP=1) against the unfused bf16 chain on the same inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from litepose_amd import arch_zoo, config
from litepose_amd.models import pose_mobilenet
from oracle import synth
arch = arch_zoo.get('search-XS')
sd = synth.make_state_dict(arch, seed=1234)
m = pose_mobilenet.get_pose_net(config.get_cfg(), is_train=False, cfg_arch=arch, storage='bf16')
m.load_state_dict(sd, strict=True)
x = synth.make_images(8, 256, seed=3).cuda()
res = {}
for mode in ('0', '1'):
    os.environ['LP_DWTP'] = mode
    m.set_profiling(True)
    out = [o.clone() for o in m(x)]
    prof = m.profile()
    m.set_profiling(False)
    taps = {k: m.tap(k).clone() for k in ('stage.0.1', 'stage.0.5', 'stage.1.1', 'stage.1.7')}
    fam = {}
    for name, ms, by, fl in prof:
        k = name.split('|')[1] if '|' in name else name
        fam[k] = fam.get(k, 0.0) + ms
    res[mode] = (out, taps, fam)
    print('LP_DWTP=%s' % mode, {k: round(v, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:6]})
for k in res['0'][1]:
    a, b = res['0'][1][k].float(), res['1'][1][k].float()
    print(k, 'max abs diff %.4g  (max |ref| %.3g)  differing %.4f' % (float((a - b).abs().max()), float(a.abs().max()), float((a != b).float().mean())))
for i in range(2):
    a, b = res['0'][0][i], res['1'][0][i]
    print('out%d max abs diff %.4g (max |ref| %.3g)' % (i, float((a - b).abs().max()), float(a.abs().max())))
