#!/usr/bin/env python
"""Times lp_parse_mid on saturated heatmaps (every pixel of every plane a plateau above the threshold: each band's key
segment overflows and takes the exact fallback) against an ordinary blob scene of the same shape (GPU box)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from litepose_amd import _native as nv, config
from litepose_amd.core import group
from oracle import synth


def run(mid_np, reps=5):
    lib = nv.lib()
    N, _, J, h1, w1 = mid_np.shape
    T, pcap = 2, 30
    cfg = config.get_cfg('crowd_pose')
    p = group.HeatmapParser(cfg, person_capacity=pcap)
    mid = torch.from_numpy(mid_np).cuda()
    need = int(lib.lp_parse_workspace_bytes(N, J, p.params.max_num_people, T, pcap))
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    ans = torch.zeros((N, pcap, J, 3 + T), device='cuda')
    cnt = torch.zeros((N,), dtype=torch.int32, device='cuda')
    sc = torch.zeros((N, pcap), device='cuda')
    f = lambda: nv.check(lib.lp_parse_mid(nv.dptr(mid), N, J, h1, w1, T, C.byref(p._q), pcap, 1, 1, nv.dptr(ans), nv.dptr(cnt),
                                          nv.dptr(sc), nv.dptr(ws), need, nv.stream_ptr()), 'lp_parse_mid')
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, int(cnt.sum())


if __name__ == '__main__':
    N, J, h1, w1 = 64, 14, 128, 128
    rng = np.random.default_rng(5)
    blob = np.zeros((N, 4, J, h1, w1), np.float32)
    for n in range(N):
        d, t = synth.blob_scene(rng, J, h1, w1, 2, n_people=8, sigma=2.0)
        blob[n, 0] = d; blob[n, 1] = d * np.float32(0.97); blob[n, 2] = t[..., 0]; blob[n, 3] = t[..., 1]
    sat = blob.copy()
    sat[:, 0] = 0.5; sat[:, 1] = 0.5                           # det == 0.5 on every pixel of every plane
    ramp = blob.copy()                                         # every pixel above the threshold AND rising with the index
    ramp[:, 0] = ramp[:, 1] = (0.2 + 0.7 * np.arange(h1 * w1, dtype=np.float32) / (h1 * w1)).reshape(h1, w1)
    stripes = blob.copy()                                      # plateau rows whose value rises down the plane: every survivor
    rows = np.where(np.arange(h1) % 3 == 0, 0.3 + 0.5 * np.arange(h1) / h1, 0.15).astype(np.float32)   # beats the M-th best so far
    stripes[:, 0] = stripes[:, 1] = rows[:, None]
    for name, m in (('blob', blob), ('saturated', sat), ('ramp', ramp), ('stripes', stripes)):
        ms, c = run(m)
        print('%-10s lp_parse_mid %8.3f ms per %d images (%d persons)' % (name, ms, N, c))
