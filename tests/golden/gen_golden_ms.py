#!/usr/bin/env python
"""Golden fixture for multi-scale test-time aggregation (valid.py:207-225), produced by
running the REAL reference `get_multi_stage_outputs` + `aggregate_results` on CPU.

    python tests/golden/gen_golden_ms.py        # build container only (needs /root/reference)

A stub "model" returns seeded synthetic low-resolution network outputs per input size, so the
fixture pins the aggregation arithmetic only (stage upsample, flip, FLIP_CONFIG permutation,
projection to the base size, sum over scales, /len(SCALE_FACTOR), tags from scale 1).  Only
outputs are stored (tests/golden/golden_ms.npz); the oracle restatement is checked on the way.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from oracle import inference_ref  # noqa: E402


def synth_outputs(seed, N, J, size):
    g = torch.Generator().manual_seed(seed)
    w, h = size
    o0 = torch.randn(N, 2 * J, h // 4, w // 4, generator=g) * 0.3
    o1 = torch.randn(N, J, h // 2, w // 2, generator=g) * 0.3
    return [o0, o1]


def main():
    torch.set_num_threads(1)
    inf, _, _ = gg.load_reference()
    out = {}
    cases = (('sq', 14, 'crowd_pose_kpt', [2, 1, 0.5], (32, 32), {2: (64, 64), 1: (32, 32), 0.5: (16, 16)}, True),
             ('rect', 17, 'coco_kpt', [1, 0.5], (48, 32), {1: (48, 32), 0.5: (24, 16)}, True),
             ('noflip', 14, 'crowd_pose_kpt', [1.5, 1], (32, 32), {1.5: (48, 48), 1: (32, 32)}, False),
             # DATASET.WITH_CENTER: NUM_JOINTS counts the centre joint (default.py:175); IGNORE_CENTER drops it
             ('center', 15, 'crowd_pose_kpt', [1], (32, 32), {1: (32, 32)}, True),
             ('centerkeep', 18, 'coco_kpt', [1], (32, 32), {1: (32, 32)}, True),
             # TEST.PROJECT2IMAGE = False (round 6; inference.py:152 false arm, :180-189, :201-206): maps stay at the stage-1
             # resolution of their scale; the tags of scale 1 and the flip-averaged heatmaps of the later scales are resized
             # to the FIRST scale's maps by aggregate_results
             ('nop2i', 14, 'crowd_pose_kpt', [2, 1, 0.5], (32, 32), {2: (64, 64), 1: (32, 32), 0.5: (16, 16)}, True),
             ('nop2i_noflip', 17, 'coco_kpt', [1.5, 1], (32, 32), {1.5: (48, 48), 1: (32, 32)}, False),
             ('nop2i_first', 14, 'crowd_pose_kpt', [1, 0.5], (48, 32), {1: (48, 32), 0.5: (24, 16)}, True))
    for name, J, ds, scales, base, sizes, flip in cases:
        cfg = gg.make_cfg(J=J, dataset=ds, input_size=base[0])
        cfg.TEST.SCALE_FACTOR = scales
        cfg.TEST.FLIP_TEST = flip
        cfg.DATASET.WITH_CENTER = name.startswith('center')
        cfg.TEST.IGNORE_CENTER = name != 'centerkeep'
        cfg.TEST.PROJECT2IMAGE = not name.startswith('nop2i')
        N = 2 if name == 'sq' else 1
        store = {}
        for idx, s in enumerate(sorted(scales, reverse=True)):
            store[s] = (synth_outputs(1000 + idx, N, J, sizes[s]), synth_outputs(2000 + idx, N, J, sizes[s]))

        class Model(object):      # model(image) / model(flip(image)) -> stored outputs of that size
            def __init__(self):
                self.calls = {}

            def __call__(self, image):
                s = [k for k, v in sizes.items() if v == (image.shape[3], image.shape[2])][0]
                i = self.calls.get(s, 0)
                self.calls[s] = i + 1
                return [t.clone() for t in store[s][i]]

        model = Model()
        final = None
        tags_list = []
        with torch.no_grad():
            for s in sorted(scales, reverse=True):
                img = torch.zeros(N, 3, sizes[s][1], sizes[s][0])
                _, hm, tg = inf.get_multi_stage_outputs(cfg, model, img, flip, cfg.TEST.PROJECT2IMAGE, base)   # valid.py:213-216
                final, tags_list = inf.aggregate_results(cfg, s, final, tags_list, hm, tg)
            final = final / float(len(scales))
            tags = torch.cat(tags_list, dim=4)
            tc = inference_ref.TestCfg(num_joints=J, dataset=ds, flip_test=flip,
                                       with_center=cfg.DATASET.WITH_CENTER, ignore_center=cfg.TEST.IGNORE_CENTER,
                                       project2image=cfg.TEST.PROJECT2IMAGE)
            ofinal, otags = inference_ref.merge_multiscale(
                [(s, store[s][0], store[s][1] if flip else None) for s in scales], tc, base)
        assert torch.equal(final, ofinal) and torch.equal(tags, otags), 'oracle != reference (%s)' % name
        out[name + '_final'] = final.numpy()
        out[name + '_tags'] = tags.numpy()
        out[name + '_meta'] = np.array([J, base[0], base[1], int(flip), N, int(cfg.DATASET.WITH_CENTER),
                                        int(cfg.TEST.IGNORE_CENTER), int(cfg.TEST.PROJECT2IMAGE)], np.int32)
        out[name + '_scales'] = np.array(sorted(scales, reverse=True), np.float64)
        for idx, s in enumerate(sorted(scales, reverse=True)):
            for f in range(2 if flip else 1):
                for k in range(2):
                    out['%s_s%d_f%d_o%d' % (name, idx, f, k)] = store[s][f][k].numpy()
        print(name, 'final', tuple(final.shape), 'tags', tuple(tags.shape), 'oracle bit-identical')
    np.savez_compressed(os.path.join(HERE, 'golden_ms.npz'), **out)
    print('wrote golden_ms.npz', os.path.getsize(os.path.join(HERE, 'golden_ms.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
