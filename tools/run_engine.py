#!/usr/bin/env python
"""Run the bench's engine configuration (XS@256, 64 images + mirrored, pcap 30, synthetic scenes) a few times
on ONE stream, for rocprofv3 traces / PMC passes over every kernel of the path incl. the AE stage:
    LP_STREAMS=1 rocprofv3 --kernel-trace --stats -d out -o t -- python tools/run_engine.py --reps 3"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from litepose_amd import arch_zoo, config, engine  # noqa: E402
from oracle import inference_ref, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='search-XS')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--size', type=int, default=0)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--warmup', type=int, default=1)
ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'])
a = ap.parse_args()
arch = arch_zoo.get(a.arch)
R = a.size or arch['img_size']
cfg = config.apply_arch(config.get_cfg(), arch)
sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30, pipeline_halves=False, storage=a.storage, options=engine.options_from_env())
x = synth.make_images(a.batch, R, seed=100).cuda()
off0, off1 = synth.lowres_offsets(200, a.batch, 14, R)
f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
# ONE stream: lp_net_forward without its internal plain/mirrored stream pair, _infer_one directly (infer_batch
# would switch the pair back on), so every kernel's trace duration is an un-shared, back-to-back launch
from litepose_amd import _native as nv  # noqa: E402
nv.check(eng._lib.lp_net_set_streams(eng.model._h, 1))
for _ in range(a.warmup + a.reps):
    out = eng._infer_one(x, offs, None, None)
torch.cuda.synchronize()
print('persons', int(out[1].sum()), 'path', eng._last[0][0])
