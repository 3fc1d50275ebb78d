#!/usr/bin/env python
"""Per-step completion times of bench.py's serving loop (GPU timestamps of each batch's completion on the
consumer stream): shows the fill / drain of the two-lane pipeline and any clock ramp after the warm-up.
    python tools/step_times.py --steps 20 --warmup 5"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from litepose_amd import arch_zoo, config, engine  # noqa: E402
from oracle import inference_ref, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warmup', type=int, default=5)
ap.add_argument('--stages', action='store_true', help='also time the NET and AE graphs of one buffer set alone')
ap.add_argument('--idle-ms', type=float, default=0.0, help='host sleep between warm-up and the timed steps')
a = ap.parse_args()
arch = arch_zoo.get('search-XS')
R, B = 256, 64
cfg = config.apply_arch(config.get_cfg(), arch)
sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30, options=engine.options_from_env())
x = synth.make_images(B, R, seed=100).cuda()
off0, off1 = synth.lowres_offsets(200, B, 14, R)
f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())


depth = eng.pipeline_depth()


def run(k, marks=None):
    pending = []

    def collect(h):
        h.result()
        h.release()
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
    for _ in range(k):
        pending.append(eng.submit(x, offsets=offs))
        if len(pending) > depth:
            collect(pending.pop(0))
    for h in pending:
        collect(h)


eng.prepare(x, offsets=offs)
run(a.warmup)
torch.cuda.synchronize()
if a.idle_ms:
    import time
    time.sleep(a.idle_ms * 1e-3)
start = torch.cuda.Event(enable_timing=True)
start.record()
marks = []
run(a.steps, marks)
torch.cuda.synchronize()
t = [start.elapsed_time(e) for e in marks]
d = np.diff([0.0] + t)
print('steps %d warmup %d: total %.3f ms = %.4f ms/step' % (a.steps, a.warmup, t[-1], t[-1] / a.steps))
print('completion intervals (ms):', ' '.join('%.2f' % v for v in d))
if a.stages and getattr(eng, '_split', False) and eng._lanes[0]['graphs']:
    ln = eng._lanes[0]
    gr = list(ln['graphs'].values())[-1]['g']
    for idx, nm in ((0, 'NET'), (1, 'AE')):
        st = ln['stream'] if idx == 0 else ln['ae_stream']
        with torch.cuda.stream(st):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr[idx].replay()
            torch.cuda.synchronize()
            e0.record(st)
            for _ in range(10):
                gr[idx].replay()
            e1.record(st)
            torch.cuda.synchronize()
            print('%s graph alone: %.3f ms' % (nm, e0.elapsed_time(e1) / 10))
