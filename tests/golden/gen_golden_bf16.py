#!/usr/bin/env python
"""A reference-held yardstick for the bf16 storage path (SURVEY.md 8 row g; VERDICT r04 weak #2: the per-launch parity of
that path was against the builder's own emulation only).

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden_bf16.py

The reference's reduced-precision evaluation is `valid.py:152-153` -> `lib/fp16_utils/fp16util.py:87-91`
(`network_to_half`: input cast, every module in half, BatchNorm layers back in float, outputs cast to float).  This script
imports the REAL reference module and the REAL `BN_convert_float` by path and applies exactly that recipe with
torch.bfloat16 in place of torch.float16 (the north star names bf16; the recipe is dtype-agnostic), on CPU, to the
seeded synthetic inputs of oracle/synth.py.  Stored: samples (every 7th value) and the max / rms of both stage outputs
of the reference in fp32 and in "half = bf16" mode -- OUTPUTS only, nothing of the reference's code.

What the fixture pins (tests/test_bf16_cpu.py on CPU, tests/test_gpu_bf16.py on the device): the bf16-storage network of
this repo (BN folded, weights and stored activations rounded once, fp32 accumulation) is a DIFFERENT bf16 realisation
of the same function than the reference's (bf16 convolutions with bf16 outputs, float BN, bf16 again): they cannot
agree bit for bit, and a 40-layer trunk amplifies every rounding choice.  So the statement is a budget in the
reference's own unit: our distance from the reference's fp32 outputs must not exceed the reference's OWN bf16 mode's
distance from them by more than a stated factor, and the two bf16 realisations must be as close to each other as each
is to fp32.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from oracle import net_ref, synth  # noqa: E402

STRIDE = 7
CASES = [('search-XS', 128, 2), ('search-XS', 256, 1), ('search-S', 224, 1), ('search-M', 256, 1)]


def load_fp16util():
    spec = importlib.util.spec_from_file_location('ref_fp16util', os.path.join(gg.REF, 'lib', 'fp16_utils', 'fp16util.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    torch.set_num_threads(8)
    _, _, pm = gg.load_reference()
    fu = load_fp16util()
    out = {}
    for arch_name, R, N in CASES:
        arch = json.load(open(os.path.join(gg.REF, 'mobile_configs', arch_name + '.json')))
        sd = synth.make_state_dict(arch, seed=1234)
        x = synth.make_images(N, R, seed=21)
        model = pm.get_pose_net(gg.make_cfg(input_size=R), is_train=False, cfg_arch=arch).eval()
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            ref32 = model(x)
            # network_to_half's recipe (fp16util.py:87-91) with bfloat16: modules in bf16, BatchNorm back in float
            half = fu.BN_convert_float(model.to(torch.bfloat16))
            refh = [o.float() for o in half(x.to(torch.bfloat16))]
            emu = net_ref.forward_bf16(x, sd, arch)
            ora = net_ref.forward(x, sd, arch)
        for k in range(2):
            assert torch.equal(ref32[k], ora[k]), 'net_ref is not bit-identical to the reference module'
            key = '%s_%d_out%d' % (arch_name, R, k)
            a32, ah, ae = ref32[k].numpy(), refh[k].numpy(), emu[k].numpy()
            out[key + '_ref32'] = a32.reshape(-1)[::STRIDE].copy()
            out[key + '_refbf16'] = ah.reshape(-1)[::STRIDE].copy()
            out[key + '_shape'] = np.array(a32.shape)
            d_ref = np.abs(ah - a32)
            d_emu = np.abs(ae - a32)
            d_x = np.abs(ae - ah)
            out[key + '_stats'] = np.array([np.abs(a32).max(), d_ref.max(), np.sqrt((d_ref ** 2).mean()),
                                            d_emu.max(), np.sqrt((d_emu ** 2).mean()), d_x.max(), np.sqrt((d_x ** 2).mean())])
            print('%-10s %3d out%d  |ref32|max %.3f   ref-bf16 vs ref32: max %.4f rms %.5f   emulation vs ref32: max %.4f rms %.5f'
                  '   emulation vs ref-bf16: max %.4f rms %.5f' % ((arch_name, R, k) + tuple(out[key + '_stats'])))
    path = os.path.join(HERE, 'golden_bf16.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
