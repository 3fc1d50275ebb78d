#!/usr/bin/env python
"""Time the AE-stage entry points separately on synthetic maps (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from litepose_amd import config, _native as nv
from litepose_amd.core import group
from oracle import synth

N = int(os.environ.get('N', 64))
det, tag = synth.blob_batch(5, 8, H=256, W=256)
det = torch.from_numpy(np.tile(det, (N // 8, 1, 1, 1))).cuda()
tag = torch.from_numpy(np.tile(tag, (N // 8, 1, 1, 1, 1))).cuda()
cfg = config.get_cfg()
lib = nv.lib()

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

for M in (30, 8, 1):
    cfg.DATASET.MAX_NUM_PEOPLE = M
    p = group.HeatmapParser(cfg, person_capacity=30)
    J, T = 14, 2
    val_k = torch.empty((N, J, M), device='cuda'); ind_k = torch.empty((N, J, M), dtype=torch.int32, device='cuda')
    tag_k = torch.empty((N, J, M, T), device='cuda')
    f = lambda: lib.lp_peaks_topk(nv.dptr(det), nv.dptr(tag), N, J, 256, 256, T, C.byref(p._q), nv.dptr(val_k), nv.dptr(ind_k), nv.dptr(tag_k), nv.stream_ptr())
    print('peaks_topk M=%d: %.3f ms' % (M, timeit(f)))
cfg.DATASET.MAX_NUM_PEOPLE = 30
p = group.HeatmapParser(cfg, person_capacity=30)
print('parse (all): %.3f ms' % timeit(lambda: p.parse_batch_device(det, tag)))

# TTA merge (tta_stage + tta_project), 64 images + 64 mirrored, XS@256 shapes
from litepose_amd.core import inference
cfg2 = config.get_cfg()
o0 = torch.randn(N, 28, 64, 64, device='cuda'); o1 = torch.randn(N, 14, 128, 128, device='cuda')
o0f = torch.randn(N, 28, 64, 64, device='cuda'); o1f = torch.randn(N, 14, 128, 128, device='cuda')
dbuf = torch.empty((N, 14, 256, 256), device='cuda'); tbuf = torch.empty((N, 14, 256, 256, 2), device='cuda')
print('tta_merge N=%d: %.3f ms' % (N, timeit(lambda: inference.tta_merge(cfg2, [o0, o1], [o0f, o1f], (256, 256), det=dbuf, tag=tbuf))))
