#!/usr/bin/env python
"""Per-launch HIP-event timing of one network forward (GPU box):
    python tools/profile_ops.py [--arch search-XS] [--batch 64] [--size 256] [--reps 5]
Prints every op with ms, algorithmic GB/s and TFLOP/s, plus per-family totals."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from litepose_amd import arch_zoo, config  # noqa: E402
from litepose_amd.models import pose_mobilenet  # noqa: E402
from oracle import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='search-XS')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--size', type=int, default=0)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--flip', type=int, default=2)
ap.add_argument('--all', action='store_true')
ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'])
ap.add_argument('--opt', action='append', default=[], help='kernel-family switch key=value (lp_net_set_option), repeatable')
a = ap.parse_args()
arch = arch_zoo.get(a.arch)
R = a.size or arch['img_size']
cfg = config.get_cfg()
m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch, storage=a.storage)
m.load_state_dict(synth.make_state_dict(arch), strict=True)
for kv in a.opt:
    k, v = kv.split('=')
    m.set_option(k, int(v))
x = synth.make_images(a.batch, R).cuda()
for _ in range(2):
    m.forward_native(x, a.flip)
m.set_profiling(True)
acc = None
for _ in range(a.reps):
    m.forward_native(x, a.flip)
    p = m.profile()
    if acc is None:
        acc = [[n, 0.0, b, f] for n, _, b, f in p]
    for i, (_, ms, _, _) in enumerate(p):
        acc[i][1] += ms / a.reps
fam = {}
tot = 0.0
for n, ms, b, f in acc:
    tot += ms
    key = n.split('|')[1] if '|' in n else n
    k = fam.setdefault(key, [0.0, 0, 0])
    k[0] += ms; k[1] += b; k[2] += f
    if a.all:
        ideal = max(b / 5.5e12, f / 120e12) * 1e3
        print('%-40s %8.4f ms %8.1f GB/s %7.2f TF   ideal %7.4f ms  x%.1f' % (n, ms, b / ms / 1e6, f / ms / 1e9, ideal, ms / ideal))
print('---- families (batch %d x flip -> %d images, %s@%d)' % (a.batch, a.batch * (2 if a.flip == 2 else 1), a.arch, R))
for k, (ms, b, f) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print('%-10s %8.4f ms %8.1f GB/s %7.2f TF' % (k, ms, b / ms / 1e6, f / ms / 1e9))
print('total %.4f ms' % tot)
