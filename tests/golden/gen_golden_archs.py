#!/usr/bin/env python
"""Network outputs of the REAL reference module for ALL seven published architectures
(mobile_configs/{search-XS,S,M,L, prune-S,M,L}.json), the ones gen_golden.py does not cover in full.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden_archs.py

Imports the reference by path exactly like gen_golden.py (nothing is copied), feeds it the seeded synthetic
inputs of oracle/synth.py and stores OUTPUT samples only: every 13th value of both stage outputs plus four
whole-tensor sums per output, for one 64x64 and (search-M / search-L, the BASELINE config-4/5 families) one
96x160 image.  While generating, oracle/net_ref.py is asserted bit-identical to the reference module and the
arch tables of litepose_amd/arch_zoo.py equal to the reference JSON files.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (puts the repo root on sys.path, loads oracle.*)
from oracle import net_ref, spec, synth  # noqa: E402

ARCHS = ['search-XS', 'search-S', 'search-M', 'search-L', 'prune-S', 'prune-M', 'prune-L']
STRIDE = 13


def stats(t):
    a = t.numpy().astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.flat[::97].sum()])


def main():
    torch.set_num_threads(1)
    _, _, pm = gg.load_reference()
    from litepose_amd import arch_zoo
    out = {}
    for arch_name in ARCHS:
        arch = json.load(open(os.path.join(gg.REF, 'mobile_configs', arch_name + '.json')))
        assert arch_zoo.get(arch_name) == arch, 'arch_zoo table differs from the reference JSON: ' + arch_name
        sizes = [(64, 64)] + ([(96, 160)] if arch_name in ('search-M', 'search-L') else [])
        model = pm.get_pose_net(gg.make_cfg(input_size=64), is_train=False, cfg_arch=arch).eval()
        shapes = spec.state_dict_shapes(arch)
        ref_sd = model.state_dict()
        assert list(ref_sd.keys()) == list(shapes.keys()), 'state_dict key scheme/order mismatch'
        sd = synth.make_state_dict(arch, seed=1234)
        model.load_state_dict(sd, strict=True)
        for H, W in sizes:
            x = synth.make_images(1, H, seed=11, w=W)
            with torch.no_grad():
                ref_out = model(x)
                ora_out = net_ref.forward(x, sd, arch)
            for k, (a, b) in enumerate(zip(ref_out, ora_out)):
                assert torch.equal(a, b), 'net_ref is not bit-identical to the reference module'
                key = '%s_%dx%d_out%d' % (arch_name, H, W, k)
                out[key + '_sample'] = a.numpy().reshape(-1)[::STRIDE].copy()
                out[key + '_stats'] = stats(a)
                out[key + '_shape'] = np.array(a.shape)
            print(arch_name, (H, W), [tuple(o.shape) for o in ref_out], 'absmax %.4f %.4f'
                  % (float(ref_out[0].abs().max()), float(ref_out[1].abs().max())))
    path = os.path.join(HERE, 'golden_archs.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
