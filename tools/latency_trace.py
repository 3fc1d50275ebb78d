#!/usr/bin/env python
"""One small batch through submit -> result, repeated: run under `rocprofv3 --kernel-trace` and read the kernel timeline of
the last iterations with tools/latency_trace.py --db x_results.db (wall per iteration, time with no kernel running, the
longest kernels).  python tools/latency_trace.py --batch 1 [--iters 40]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--iters', type=int, default=40)
ap.add_argument('--db', default=None)
args = ap.parse_args()

if args.db:
    import sqlite3
    c = sqlite3.connect(args.db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    rows = [(n.split('(')[0].replace('void ', '').replace('lp::', ''), s, e) for n, s, e in rows]
    # iterations are separated by the host sync: gaps > 30 us with nothing running
    its, cur, hi = [], [rows[0]], rows[0][2]
    for r in rows[1:]:
        if r[1] - hi > 30e3:
            its.append(cur)
            cur = []
        cur.append(r)
        hi = max(hi, r[2])
    its.append(cur)
    its = [it for it in its if len(it) > 40][-10:]
    for it in its[-3:]:
        t0, t1 = it[0][1], max(r[2] for r in it)
        busy, last = 0, t0
        for n, s, e in sorted(it, key=lambda r: r[1]):
            if e > last:
                busy += e - max(s, last)
                last = e
        print('%d kernels, first start -> last end %.1f us, some kernel running %.1f us, idle %.1f us; sum of durations %.1f us'
              % (len(it), (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, sum(e - s for _, s, e in it) / 1e3))
    agg = {}
    for n, s, e in its[-1]:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    for n, (k, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print('  %-44s %3d launches %8.1f us' % (n[:44], k, d / 1e3))
    sys.exit(0)

import numpy as np
import torch
from litepose_amd import arch_zoo, config, engine
from oracle import inference_ref, synth

arch = arch_zoo.get('search-XS')
cfg = config.apply_arch(config.get_cfg(), arch)
sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
nb, R, J = args.batch, 256, 14
xs = synth.make_images(nb, R, seed=400 + nb).cuda()
o0, o1 = synth.lowres_offsets(500 + nb, nb, J, R)
g0, g1 = synth.flip_offsets(o0, o1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
offs = (torch.from_numpy(np.concatenate([o0, g0])).cuda(), torch.from_numpy(np.concatenate([o1, g1])).cuda())
eng.prepare(xs, offsets=offs)
import time
lat = []
for _ in range(args.iters):
    torch.cuda.synchronize()
    t = time.perf_counter()
    with eng.submit(xs, offsets=offs) as res:
        pass
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t) * 1e3)
    time.sleep(0.0005)
print('batch %d: median %.4f ms, min %.4f' % (nb, sorted(lat)[len(lat) // 2], min(lat)))
