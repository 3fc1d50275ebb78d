#!/usr/bin/env python
"""HBM read / write / copy bandwidth as seen by plain torch kernels (GPU box): calibrates the
'achievable' side of the roofline for write-heavy kernels."""
import time
import torch

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device='cuda'); y = torch.empty_like(x)
    x.normal_()
    tw = t(lambda: x.fill_(1.0)); tr = t(lambda: x.sum()); tc = t(lambda: y.copy_(x))
    ta = t(lambda: x.add_(1.0))
    print('%5d MB  write %.2f TB/s   read(sum) %.2f TB/s   copy %.2f TB/s (r+w)   rmw add_ %.2f TB/s (r+w)' % (
        mb, mb / 1048576 / tw, mb / 1048576 / tr, 2 * mb / 1048576 / tc, 2 * mb / 1048576 / ta))
