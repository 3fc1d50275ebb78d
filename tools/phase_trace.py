#!/usr/bin/env python
"""Phase trace of the fused bf16 block kernels (round 6; needs the `trace` flavour: python -m litepose_amd.build --flavour trace):
    LP_NATIVE_FLAVOUR=trace python tools/phase_trace.py --arch search-S --size 448 --batch 32 [--opt mbtq=0]
prints, per kernel and CK, the average shader-clock cycles a wave spends in each phase per launch-workgroup."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from litepose_amd import _native as nv, arch_zoo, config  # noqa: E402
from litepose_amd.models import pose_mobilenet  # noqa: E402
from oracle import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='search-S')
ap.add_argument('--size', type=int, default=0)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--opt', action='append', default=[])
a = ap.parse_args()
arch = arch_zoo.get(a.arch)
R = a.size or arch['img_size']
m = pose_mobilenet.get_pose_net(config.get_cfg(), cfg_arch=arch, storage='bf16')
m.load_state_dict(synth.make_state_dict(arch), strict=True)
for kv in a.opt:
    k, v = kv.split('=')
    m.set_option(k, int(v))
x = synth.make_images(a.batch, R).cuda()
m.forward_native(x, 2)
torch.cuda.synchronize()
tab = (C.c_uint64 * 128)()
nv.check(nv.lib().lp_phase_trace_read(tab, 1), 'lp_phase_trace_read')
REPS = 3
for _ in range(REPS):
    m.forward_native(x, 2)
torch.cuda.synchronize()
nv.check(nv.lib().lp_phase_trace_read(tab, 1), 'lp_phase_trace_read')
names = ['prologue', 'depthwise', 'drain+bar', 'project', 'expand', 'barrier', 'epilogue']
print('%s@%d batch %d (x2 mirrored), %s: average shader-clock cycles per wave and workgroup' % (a.arch, R, a.batch, a.opt))
print('%-6s %-3s %9s ' % ('kernel', 'CK', 'waves') + ' '.join('%10s' % n for n in names) + '      total')
for w, kn in enumerate(('mbtb', 'mbtq')):
    for ck in range(8):
        row = [int(tab[(w * 8 + ck) * 8 + k]) for k in range(8)]
        if not row[7]:
            continue
        avg = [row[k] / row[7] for k in range(7)]
        print('%-6s %-3d %9d ' % (kn, ck + 1, row[7] // REPS) + ' '.join('%10.0f' % v for v in avg) + ' %10.0f' % sum(avg))
