#!/usr/bin/env python
"""Two kernel-family configurations of the device network (lp_net_set_option) against each other: every fused block's
output tensor and both network outputs compared bit for bit.  Round 6: the persistent mbtbp_kernel and the 4-wave mbtq_kernel
against round 3's one-tile-per-workgroup mbtb_kernel.
    python tools/block_ab_check.py --a mbtb=2,mbtq=0 --b mbtb=1,mbtq=0 [--arch search-S] [--size 448] [--batch 4] [--storage bf16]"""
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
ap.add_argument('--storage', default='bf16')
ap.add_argument('--a', default='mbtb=2,mbtq=0')
ap.add_argument('--b', default='mbtb=1,mbtq=0')
a = ap.parse_args()
arch = arch_zoo.get(a.arch)
R = a.size or arch['img_size']
cfg = config.get_cfg()
m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch, storage=a.storage)
m.load_state_dict(synth.make_state_dict(arch), strict=True)
x = synth.make_images(a.batch, R).cuda()
res = {}
for mode, opts in ((0, a.a), (2, a.b)):
    for kv in opts.split(','):
        k, v = kv.split('=')
        m.set_option(k, int(v))
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
