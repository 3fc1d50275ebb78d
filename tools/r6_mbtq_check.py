#!/usr/bin/env python
"""Round 6: mbtq_kernel (4-wave workgroups, two per CU) against mbtb_kernel on the device: every block boundary of the
network compared bit for bit, on ragged / bordered planes too.
    python tools/r6_mbtq_check.py [--arch search-S] [--size 448] [--batch 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from litepose_amd import arch_zoo, config  # noqa: E402
from litepose_amd.models import pose_mobilenet  # noqa: E402
from oracle import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='search-S')
ap.add_argument('--size', type=int, default=0)
ap.add_argument('--batch', type=int, default=4)
a = ap.parse_args()
arch = arch_zoo.get(a.arch)
R = a.size or arch['img_size']
cfg = config.get_cfg()
m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch, storage='bf16')
m.load_state_dict(synth.make_state_dict(arch), strict=True)
x = synth.make_images(a.batch, R).cuda()
res = {}
for mode in (0, 2):
    m.set_option('mbtq', mode)
    m.set_profiling(True)
    outs = [o.clone() for o in m.forward_native(x, 2)]
    torch.cuda.synchronize()
    prof = m.profile()
    m.set_profiling(False)
    names = [n.split('|')[0].split('.inv')[0] for n, *_ in prof if 'mbtq' in n or 'mbtb_kernel' in n]
    kern = {n.split('|')[0].split('.inv')[0]: n.split('|')[1] for n, *_ in prof if '+point_conv' in n}
    taps = {n: m.tap(n + '.point_conv').clone() for n in kern}
    res[mode] = (outs, taps, kern)
bad = 0
for n, t0 in res[0][1].items():
    t2 = res[2][1][n]
    d = t0 != t2
    nd = int(d.sum())
    if res[2][2][n] != res[0][2][n] or nd:
        print('%-12s %-16s vs %-16s  differing elements %d of %d' % (n, res[0][2][n], res[2][2][n], nd, t0.numel()))
    bad += nd
for o0, o2 in zip(res[0][0], res[2][0]):
    print('output max abs diff', float((o0 - o2).abs().max()))
print('TOTAL differing tap elements:', bad)
