#!/usr/bin/env python
"""mb16p_kernel (LP_MB16=5) against mb16_kernel (1) and the unfused chain (0): every stage-3/4 block tap bitwise, then
the per-launch time of the 19 blocks of XS@256 at the bench batch (64 images + mirrored) for both fused forms.
    python tools/mb16p_check.py [--archs search-XS,search-S,search-L]"""
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
ap.add_argument('--modes', default='5,1')
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--dbg', default='')
ap.add_argument('--dbg1', default='')
a = ap.parse_args()
cfg = config.get_cfg()
bad = 0
for name in a.archs.split(','):
    arch = arch_zoo.get(name)
    m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch)
    m.load_state_dict(synth.make_state_dict(arch), strict=True)
    x = synth.make_images(5, 256, seed=31).cuda()
    names = ['stage.%d.%d' % (s, b) for s in (2, 3) for b in range(10)]
    res = {}
    for mode in a.modes.split(',') + ['0']:
        os.environ['LP_MB16'] = mode
        m.set_profiling(True)
        out = [o.clone() for o in m(x)]
        kern = [n.split('|')[1] for n, _, _, _ in m.profile()]
        m.set_profiling(False)
        res[mode] = (out, {k: m.tap(k).clone() for k in names}, kern)
    for mode in a.modes.split(','):
        diff = [k for k in names if not torch.equal(res[mode][1][k], res['0'][1][k])]
        worst = max(float((res[mode][1][k] - res['0'][1][k]).abs().max()) for k in names)
        outd = max(float((p - q).abs().max()) for p, q in zip(res[mode][0], res['0'][0]))
        nk = sum(1 for k in res[mode][2] if k.startswith('mb16'))
        print('%s LP_MB16=%s: %d fused launches (%s), taps not bitwise equal to the unfused chain: %s, worst tap diff %.3g, '
              'output diff %.3g' % (name, mode, nk, sorted(set(k for k in res[mode][2] if k.startswith('mb16'))), diff, worst, outd))
        if name == 'search-XS' and (diff or outd != 0.0):
            bad += 1
# timing at the bench batch
arch = arch_zoo.get('search-XS')
m = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch)
m.load_state_dict(synth.make_state_dict(arch), strict=True)
x = synth.make_images(64, 256).cuda()
for mode in a.modes.split(','):
    os.environ['LP_MB16'] = mode
    for _ in range(2):
        m.forward_native(x, 2)
    m.set_profiling(True)
    tot = {}
    for _ in range(a.reps):
        m.forward_native(x, 2)
        for n, ms, b, f in m.profile():
            k = n.split('|')[1]
            tot[k] = tot.get(k, 0.0) + ms / a.reps
    last = m.profile()
    m.set_profiling(False)
    print('LP_MB16=%s XS@256 128 images: %s   network total %.4f ms' % (
        mode, ', '.join('%s %.4f ms' % kv for kv in sorted(tot.items(), key=lambda kv: -kv[1])[:4]), sum(tot.values())))
    for n, ms, b, f in last:
        if 'stage.2.1.' in n or 'stage.3.0.' in n or 'stage.3.1.' in n:
            print('   %-50s %.4f ms' % (n, ms))
if a.dbg:
    os.environ['LP_MB16'] = '5'
    for dbg in a.dbg.split(','):
        os.environ['LP_MB16P_DBG'] = dbg
        for _ in range(2):
            m.forward_native(x, 2)
        m.set_profiling(True)
        t = 0.0
        for _ in range(a.reps):
            m.forward_native(x, 2)
            t += sum(ms for n, ms, b, f in m.profile() if 'stage.3.5.' in n) / a.reps
        m.set_profiling(False)
        print('LP_MB16P_DBG=%s: stage.3.5 (80 -> 480 -> 80) %.4f ms' % (dbg, t))
    os.environ.pop('LP_MB16P_DBG', None)
if a.dbg1:
    os.environ['LP_MB16'] = '1'
    for dbg in ['0'] + a.dbg1.split(','):
        os.environ['LP_MB16_DBG'] = dbg
        for _ in range(2):
            m.forward_native(x, 2)
        m.set_profiling(True)
        t = 0.0
        for _ in range(a.reps):
            m.forward_native(x, 2)
            t += sum(ms for n, ms, b, f in m.profile() if 'stage.3.5.' in n) / a.reps
        m.set_profiling(False)
        print('LP_MB16_DBG=%s (mb16_kernel): stage.3.5 (80 -> 480 -> 80) %.4f ms' % (dbg, t))
    os.environ.pop('LP_MB16_DBG', None)
os.environ.pop('LP_MB16', None)
print('MB16P_CHECK', 'FAIL' if bad else 'OK')
sys.exit(1 if bad else 0)
