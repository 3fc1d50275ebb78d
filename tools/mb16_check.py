#!/usr/bin/env python
"""mb16_kernel: a run of residual blocks per launch (option mb16_run = 1, default) against one block per launch
(mb16_run = 0) and the unfused chain (mb16 = 0): every stage-3 / stage-4 block tap and both outputs BITWISE, then the
per-launch times of the three forms at the bench batch (64 images + mirrored).
    python tools/mb16_check.py [--archs search-XS,search-S,search-L]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from litepose_amd import arch_zoo, config  # noqa: E402
from litepose_amd.models import pose_mobilenet  # noqa: E402
from oracle import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--archs', default='search-XS,search-S,search-L')
ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
cfg = config.get_cfg()
MODES = {'run': (1, 1), 'block': (1, 0), 'chain': (0, 0)}
bad = 0
for name in a.archs.split(','):
    arch = arch_zoo.get(name)
    m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch)
    m.load_state_dict(synth.make_state_dict(arch), strict=True)
    x = synth.make_images(5, 256, seed=31).cuda()
    names = ['stage.%d.%d' % (s, b) for s in (2, 3) for b in range(10)]
    res = {}
    for mode, (mb16, run) in MODES.items():
        m.set_option('mb16', mb16); m.set_option('mb16_run', run); m.set_option('mb16_min', 0)
        m.set_profiling(True)
        out = [o.clone() for o in m(x)]
        kern = [n for n, _, _, _ in m.profile()]
        m.set_profiling(False)
        res[mode] = (out, {k: m.tap(k).clone() for k in names}, kern)
    for mode in ('run', 'block'):
        diff = [k for k in names if not torch.equal(res[mode][1][k], res['chain'][1][k])]
        outd = max(float((p - q).abs().max()) for p, q in zip(res[mode][0], res['chain'][0]))
        launches = [k for k in res[mode][2] if k.endswith('mb16_kernel')]
        print('%s %s: %d mb16 launches %s; taps not bitwise equal to the unfused chain: %s, output diff %.3g' % (
            name, mode, len(launches), [k.split('.inv')[0] for k in launches], diff, outd))
        if diff or outd != 0.0:
            bad += 1
arch = arch_zoo.get('search-XS')
m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch)
m.load_state_dict(synth.make_state_dict(arch), strict=True)
x = synth.make_images(64, 256).cuda()
for mode in ('run', 'block'):
    mb16, run = MODES[mode]
    m.set_option('mb16', mb16); m.set_option('mb16_run', run)
    for _ in range(2):
        m.forward_native(x, 2)
    m.set_profiling(True)
    tot = {}
    per = {}
    for _ in range(a.reps):
        m.forward_native(x, 2)
        for n, ms, b, f in m.profile():
            k = n.split('|')[1]
            tot[k] = tot.get(k, 0.0) + ms / a.reps
            if k == 'mb16_kernel':
                per[n] = per.get(n, 0.0) + ms / a.reps
    m.set_profiling(False)
    print('%s XS@256 128 images: %s   network total %.4f ms' % (
        mode, ', '.join('%s %.4f ms' % kv for kv in sorted(tot.items(), key=lambda kv: -kv[1])[:4]), sum(tot.values())))
    if mode == 'run':
        for n, ms in per.items():
            print('   %-50s %.4f ms' % (n, ms))
print('MB16_CHECK', 'FAIL' if bad else 'OK')
sys.exit(1 if bad else 0)
