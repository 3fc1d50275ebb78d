#!/usr/bin/env python
"""Per-launch report of the bf16-storage network against oracle.net_ref.bf16_plan fed the device's own inputs
(the table behind tests/test_gpu_bf16.py::test_bf16_every_launch_vs_emulation_on_device_inputs), plus HIP-event
time per launch.   python tools/bf16_diag.py --arch search-S --size 448 --batch 2 [--time-batch 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import synth  # noqa: E402
from tests.test_gpu_bf16 import _model, layerwise_report  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='search-XS')
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--time-batch', type=int, default=0, help='also time every launch at this batch (x2 flip)')
    ap.add_argument('--only-bad', action='store_true')
    a = ap.parse_args()
    m, arch, sd = _model(a.arch)
    x = synth.make_images(a.batch, a.size, seed=41)
    rows = layerwise_report(m, arch, sd, x)
    nbad = 0
    for name, dmax, ulps, frac, head in rows:
        bad = (dmax > 2e-5) if head else (ulps > 1.0 or frac > 0.02)
        nbad += bad
        if bad or not a.only_bad:
            print('%-28s max|d| %.3e  %.2f ulp  differ %.4f %s' % (name, dmax, ulps, frac, 'BAD' if bad else ''))
    print('%s@%d N=%d: %d launches, %d bad' % (a.arch, a.size, a.batch, len(rows), nbad))
    if a.time_batch:
        xb = synth.make_images(a.time_batch, a.size, seed=1).cuda()
        m.forward_native(xb, 2)
        m.set_profiling(True)
        agg, tot = {}, 0.0
        reps = 3
        for _ in range(reps):
            m.forward_native(xb, 2)
            for name, ms, by, fl in m.profile():
                print('%-44s %.4f ms  %7.1f GB/s  %6.2f TF' % (name, ms, by / ms / 1e6, fl / ms / 1e9)) if _ == reps - 1 else None
                fam = name.split('|')[1]
                g = agg.setdefault(fam, [0.0, 0, 0])
                g[0] += ms / reps
                g[1] += by / reps
                g[2] += fl / reps
                tot += ms / reps
        m.set_profiling(False)
        print('---- families (%d images x flip)' % a.time_batch)
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print('%-20s %.4f ms  %7.1f GB/s  %6.2f TF' % (k, v[0], v[1] / v[0] / 1e6, v[2] / v[0] / 1e9))
        print('total %.4f ms' % tot)


if __name__ == '__main__':
    main()
