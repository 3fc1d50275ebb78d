#!/usr/bin/env python
"""HIP-event time of the two merge kernels of the default AE path on the bench shapes (64 images + mirrored, XS@256):
stage merge with additive maps (lp_tta_stage_add) and the det-only projection (lp_tta_project, tag = NULL)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from litepose_amd import config  # noqa: E402
from litepose_amd.core import inference  # noqa: E402

N = int(os.environ.get('N', 64))
cfg = config.get_cfg()
o0 = torch.randn(2 * N, 28, 64, 64, device='cuda')
o1 = torch.randn(2 * N, 14, 128, 128, device='cuda')
a0, a1 = torch.randn_like(o0), torch.randn_like(o1)
mid = torch.empty(N * 4 * 14 * 128 * 128 * 4 + 1024, dtype=torch.uint8, device='cuda')
det = torch.empty((N, 14, 256, 256), device='cuda')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


outs, outs_f = [o0[:N], o1[:N]], [o0[N:], o1[N:]]
print('tta_stage (+ additive maps) N=%d: %.1f us' % (N, timeit(lambda: inference.tta_stage(cfg, outs, outs_f, mid, add=(a0, a1)))))
print('tta_stage (plain)           N=%d: %.1f us' % (N, timeit(lambda: inference.tta_stage(cfg, outs, outs_f, mid))))
print('tta_project det only        N=%d: %.1f us' % (N, timeit(lambda: inference.tta_project(mid, N, 14, 128, 128, (256, 256), 2, det=det, det_only=True))))
