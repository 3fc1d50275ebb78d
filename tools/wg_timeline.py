#!/usr/bin/env python
"""Workgroup timeline + per-phase times of one fused bf16 block launch (`trace` flavour; round 6):
    LP_NATIVE_FLAVOUR=trace python tools/wg_timeline.py --arch search-S --size 448 --batch 32 --cexp 96 [--opt mbtb=2]
ONE stream, one launch per block (profiling mode).  Per CU (XCC, SE, CU of HW_ID): how many workgroups it ran, how long each
lived, how much of the launch it was busy, how many were resident at once; per wave: s_memtime ticks per phase."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from litepose_amd import _native as nv, arch_zoo, config  # noqa: E402
from litepose_amd.models import pose_mobilenet  # noqa: E402
from oracle import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='search-S')
ap.add_argument('--size', type=int, default=0)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--cexp', type=int, default=96)
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
lib = nv.lib()
m.set_profiling(True)                       # one launch sequence over all 2 N images, one stream
m.forward_native(x, 2)
torch.cuda.synchronize()
nv.check(lib.lp_wg_trace_read(None, 0, a.cexp), 'lp_wg_trace_read')
m.forward_native(x, 2)
torch.cuda.synchronize()
prof = m.profile()
NW = 16384
tab = (C.c_uint64 * (4 * NW))()
nv.check(lib.lp_wg_trace_read(tab, NW, 0), 'lp_wg_trace_read')
wt = (C.c_uint64 * (64 * NW))()
nv.check(lib.lp_phase_trace_read(wt, NW), 'lp_phase_trace_read')
t = np.frombuffer(tab, dtype=np.uint64).reshape(NW, 4).astype(np.int64)
w = np.frombuffer(wt, dtype=np.uint64).reshape(NW, 8, 8).astype(np.float64)
live = t[:, 0] > 0
t, w = t[live], w[live]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) * 0.01, (t[:, 2] - t0) * 0.01          # us
hw, xcc = t[:, 1] & 0xffffffff, t[:, 1] >> 32
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
life = end - start
kern = [(n, ms) for n, ms, _, _ in prof if '+point_conv' in n]
print('%s@%d batch %d, Cexp %d, %s: %d workgroups recorded (the LAST launch with that Cexp), span %.1f us'
      % (a.arch, R, a.batch, a.cexp, a.opt, len(t), end.max()))
print('launch times of the fused blocks (HIP events, traced build): ' + ' '.join('%s %.1f' % (n.split('.inv')[0], ms * 1e3) for n, ms in kern))
tpn = t[:, 3].mean() / (life.mean() * 1e3)
print('workgroup life: mean %.2f us, p10 %.2f, p50 %.2f, p90 %.2f, max %.2f; s_memtime: %.2f ticks per ns'
      % (life.mean(), np.percentile(life, 10), np.percentile(life, 50), np.percentile(life, 90), life.max(), tpn))
cus = np.unique(cu)
busy, conc, cnt, lastend = [], [], [], []
for c in cus:
    s_, e_ = start[cu == c], end[cu == c]
    ev = sorted([(v, 1) for v in s_] + [(v, -1) for v in e_])
    cur = mx = 0
    tb = 0.0
    last = 0.0
    for v, d in ev:
        if cur > 0:
            tb += v - last
        cur += d
        mx = max(mx, cur)
        last = v
    busy.append(tb)
    conc.append(mx)
    cnt.append(len(s_))
    lastend.append(e_.max())
busy, conc, cnt, lastend = np.array(busy), np.array(conc), np.array(cnt), np.array(lastend)
print('CUs seen: %d; workgroups per CU: min %d mean %.2f max %d; max resident at once on a CU: %s'
      % (len(cus), cnt.min(), cnt.mean(), cnt.max(), dict(zip(*[v.tolist() for v in np.unique(conc, return_counts=True)]))))
print('CU busy (>= 1 workgroup resident): mean %.1f us = %.0f %% of the span; average residency while busy %.2f; a CU\'s last '
      'workgroup ends at: p10 %.1f p50 %.1f p90 %.1f max %.1f us'
      % (busy.mean(), 100 * busy.mean() / end.max(), life.sum() / busy.sum(), np.percentile(lastend, 10),
         np.percentile(lastend, 50), np.percentile(lastend, 90), lastend.max()))
c0 = cus[0]
o = np.argsort(start[cu == c0])
print('first CU: start-end (us):', ' '.join('%.1f-%.1f' % (a_, b_) for a_, b_ in zip(start[cu == c0][o][:14], end[cu == c0][o][:14])))
names = ['prologue', 'depthwise', 'drain+bar', 'project', 'expand', 'barrier', 'epilogue']
wv = w[w[:, :, 7] > 0]                                             # waves that ran (mbtq: 4 per workgroup)
tiles = wv[:, 7].sum()
print('per wave and TILE, ticks (%.0f waves, %.2f tiles per wave): ' % (len(wv), wv[:, 7].mean())
      + ' '.join('%s %.0f' % (n, wv[:, k].sum() / tiles) for k, n in enumerate(names))
      + ' | total %.0f = %.2f us' % (wv[:, :7].sum() / tiles, wv[:, :7].sum() / tiles / tpn / 1e3))
