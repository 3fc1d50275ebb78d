#!/usr/bin/env python
"""Generate the golden fixtures by running the REAL reference code on CPU.

Run in the build container only (needs /root/reference and the munkres 1.1.4
copy at /opt/conda/lib/python3.9/site-packages/munkres.py):

    python tests/golden/gen_golden.py

Nothing from the reference is copied: its files are imported by path (SURVEY.md
§8c recipe), fed the seeded synthetic inputs of oracle/synth.py, and only their
OUTPUTS are stored (tests/golden/*.npz).  The GPU box has no /root/reference, so
tests read these fixtures instead.  While generating, every oracle restatement is
also checked against the reference; a mismatch aborts.
"""
import importlib.util
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'
MUNKRES = '/opt/conda/lib/python3.9/site-packages/munkres.py'

from oracle import spec, net_ref, inference_ref, group_ref, munkres_ref, synth  # noqa: E402


def _load(name, path):
    s = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(s)
    sys.modules[name] = m
    s.loader.exec_module(m)
    return m


def load_reference():
    warnings.simplefilter('ignore')
    sys.path[:0] = [REF, os.path.join(REF, 'lib')]
    _load('munkres', MUNKRES)
    ds = types.ModuleType('dataset')
    dt = types.ModuleType('dataset.transforms')
    dt.FLIP_CONFIG = inference_ref.FLIP_CONFIG   # values checked against the file below
    ds.transforms = dt
    sys.modules['dataset'] = ds
    sys.modules['dataset.transforms'] = dt
    inf = _load('ref_inference', os.path.join(REF, 'lib/core/inference.py'))
    grp = _load('ref_group', os.path.join(REF, 'lib/core/group.py'))
    pm = _load('ref_pose_mobilenet', os.path.join(REF, 'lib/models/pose_mobilenet.py'))
    # FLIP_CONFIG literal check against the reference file text
    txt = open(os.path.join(REF, 'lib/dataset/transforms/build.py')).read()
    ns = {}
    exec(txt[txt.index('FLIP_CONFIG'):txt.index('}', txt.index('FLIP_CONFIG')) + 1], ns)
    assert ns['FLIP_CONFIG'] == inference_ref.FLIP_CONFIG
    return inf, grp, pm


def make_cfg(J=14, dataset='crowd_pose_kpt', input_size=256):
    NS = types.SimpleNamespace
    return NS(
        DATASET=NS(NUM_JOINTS=J, MAX_NUM_PEOPLE=30, WITH_CENTER=False, DATASET=dataset,
                   INPUT_SIZE=input_size),
        TEST=NS(DETECTION_THRESHOLD=0.1, TAG_THRESHOLD=1.0, USE_DETECTION_VAL=True,
                IGNORE_TOO_MUCH=False, IGNORE_CENTER=True, NMS_KERNEL=5, NMS_PADDING=2,
                FLIP_TEST=True, PROJECT2IMAGE=True, SCALE_FACTOR=[1],
                WITH_HEATMAPS=(True, True), WITH_AE=(True, False), ADJUST=True, REFINE=True),
        LOSS=NS(NUM_STAGES=2, WITH_HEATMAPS_LOSS=[True, True], WITH_AE_LOSS=[True, False]),
        MODEL=NS(NAME='pose_mobilenet', NUM_JOINTS=J, TAG_PER_JOINT=True, INIT_WEIGHTS=False,
                 PRETRAINED='', EXTRA=NS(NUM_DECONV_LAYERS=3, NUM_DECONV_KERNELS=[4, 4, 4])),
    )


def ref_parse(grp, cfg, det, tag):
    """Reference parse on one image (numpy in / numpy out)."""
    parser = grp.HeatmapParser(cfg)
    d = torch.from_numpy(det[None].copy())
    t = torch.from_numpy(tag[None].copy())
    tk = parser.top_k(d, t)
    ans, scores = parser.parse(d, t, True, True)
    a = ans[0]
    if not isinstance(a, np.ndarray) or a.ndim != 3:
        a = np.zeros((0, det.shape[0], 3 + tag.shape[3]), np.float32)
    return tk, np.asarray(a, np.float32), np.asarray([float(s) for s in scores], np.float32), \
        np.asarray(scores, dtype=np.float32)


def main():
    torch.set_num_threads(1)
    inf, grp, pm = load_reference()
    out = {}

    # ---------------------------------------------------------------- network
    for arch_name, R, N in (('search-XS', 64, 2), ('search-XS', 128, 1), ('search-S', 64, 1)):
        arch = json.load(open(os.path.join(REF, 'mobile_configs', arch_name + '.json')))
        cfg = make_cfg(input_size=R)
        model = pm.get_pose_net(cfg, is_train=False, cfg_arch=arch).eval()
        ref_sd = model.state_dict()
        shapes = spec.state_dict_shapes(arch)
        assert list(ref_sd.keys()) == list(shapes.keys()), 'state_dict key scheme/order mismatch'
        for k, v in ref_sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
        sd = synth.make_state_dict(arch, seed=1234)
        model.load_state_dict(sd, strict=True)
        x = synth.make_images(N, R, seed=7)
        with torch.no_grad():
            ref_out = model(x)
            ora_out = net_ref.forward(x, sd, arch)
            for a, b in zip(ref_out, ora_out):
                assert torch.equal(a, b), 'net_ref is not bit-identical to the reference module'
            _, hm, tg = inf.get_multi_stage_outputs(cfg, model, x, True, True, (R, R))
            fh, tl = inf.aggregate_results(cfg, 1, None, [], hm, tg)
            fh = fh / 1.0
            tags = torch.cat(tl, dim=4)
            tc = inference_ref.TestCfg()
            ofh, otags = inference_ref.run(lambda im: net_ref.forward(im, sd, arch), x, tc)
            assert torch.equal(fh, ofh) and torch.equal(tags, otags), 'inference_ref mismatch'
        tag_ = '%s_%d' % (arch_name, R)
        print(tag_, 'out0', tuple(ref_out[0].shape), 'absmax', float(ref_out[0].abs().max()),
              'heat absmax', float(fh.abs().max()))
        if R == 64:
            out['net_%s_out0' % tag_] = ref_out[0].numpy()
            out['net_%s_out1' % tag_] = ref_out[1].numpy()
            out['net_%s_heat' % tag_] = fh.numpy()
            out['net_%s_tags' % tag_] = tags.numpy()
        else:
            for nm, t in (('out0', ref_out[0]), ('out1', ref_out[1]), ('heat', fh), ('tags', tags)):
                a = t.numpy().astype(np.float64)
                out['net_%s_%s_stats' % (tag_, nm)] = np.array(
                    [a.sum(), np.abs(a).sum(), (a * a).sum(), a.flat[::97].sum()])
                out['net_%s_%s_sample' % (tag_, nm)] = t.numpy().reshape(-1)[::997].copy()
        # network-driven "stress" AE case at R=128 (noise peaks above threshold)
        if R == 128:
            sd2 = synth.make_state_dict(arch, seed=1234, head_gain=6.0)
            model.load_state_dict(sd2, strict=True)
            with torch.no_grad():
                _, hm, tg = inf.get_multi_stage_outputs(cfg, model, x, True, True, (R, R))
                fh, tl = inf.aggregate_results(cfg, 1, None, [], hm, tg)
                tags = torch.cat(tl, dim=4)
            tk, a, s, _ = ref_parse(grp, cfg, fh[0].numpy(), tags[0].numpy())
            o_a, o_s = group_ref.HeatmapParser(group_ref.Params()).parse_image(fh[0].numpy(), tags[0].numpy())
            print('stress persons', a.shape, 'heat absmax', float(fh.abs().max()))
            assert a.shape == o_a.shape and np.array_equal(a, o_a) and np.array_equal(s, o_s)
            out['stress_heat'] = fh[0].numpy()
            out['stress_tags'] = tags[0].numpy()
            out['stress_ans'] = a
            out['stress_scores'] = s

    # ---------------------------------------------------------------- munkres
    rng = np.random.default_rng(11)
    import munkres as real_munkres
    mk = []
    for trial in range(300):
        r = int(rng.integers(1, 13))
        c = int(rng.integers(r, 14))          # reference always has cols >= rows
        kind = trial % 4
        if kind == 0:
            m = rng.uniform(-1, 300, size=(r, c))
        elif kind == 1:
            m = rng.integers(0, 4, size=(r, c)).astype(np.float64)          # heavy ties
        elif kind == 2:
            m = np.round(rng.uniform(0, 3, size=(r, c))) * 100 - rng.uniform(0.1, 1, size=(r, 1))
        else:
            k = int(rng.integers(1, c + 1))
            m = np.round(rng.uniform(0, 3, size=(r, c))) * 100 - rng.uniform(0.1, 1, size=(r, 1))
            m[:, k:] = 1e10
        ref_pairs = real_munkres.Munkres().compute(m.copy())
        ora_pairs = munkres_ref.compute(m.copy())
        assert [tuple(p) for p in ref_pairs] == ora_pairs, (trial, ref_pairs, ora_pairs)
        mk.append((m, np.array(ref_pairs, dtype=np.int32).reshape(-1, 2)))
    out['munkres_n'] = np.array([len(mk)])
    for i, (m, p) in enumerate(mk[:60]):
        out['munkres_m%d' % i] = m
        out['munkres_p%d' % i] = p

    # ---------------------------------------------------------------- AE parser on blob scenes
    scenes = []
    for (seed, J, R, T, people, dataset) in (
            (101, 14, 128, 2, [0, 1, 3, 6, 9, 12], 'crowd_pose_kpt'),
            (102, 14, 256, 2, [5, 35], 'crowd_pose_kpt'),
            (103, 17, 128, 1, [2, 7], 'coco_kpt'),
            (104, 14, 96, 2, [4, 8, 2, 10, 1, 7, 3, 11], 'crowd_pose_kpt')):
        cfg = make_cfg(J=J, dataset=dataset, input_size=R)
        det, tag = synth.blob_batch(seed, len(people), J=J, H=R, W=R, T=T, people=people,
                                    sigma=4.0 * R / 256.0 if R >= 128 else 2.0)
        params = group_ref.Params(num_joints=J)
        for n in range(len(people)):
            tk, a, s, _ = ref_parse(grp, cfg, det[n], tag[n])
            otk = group_ref.top_k(det[n][None], tag[n][None], params)
            # top_k: compare where the reference value is positive (tie/filler order is
            # implementation-defined in torch.topk, see group_ref docstring)
            pos = tk['val_k'][0] > 0
            assert np.array_equal(tk['val_k'][0] * pos, otk['val_k'][0])
            v = tk['val_k'][0]
            uniq = (v[:, :, None] == v[:, None, :]).sum(axis=2) == 1   # drop exact ties
            pos = pos & uniq
            assert np.array_equal(tk['loc_k'][0][pos], otk['loc_k'][0][pos])
            assert np.array_equal(tk['tag_k'][0][pos], otk['tag_k'][0][pos])
            o_a, o_s = group_ref.HeatmapParser(params).parse_image(det[n], tag[n])
            assert a.shape == o_a.shape, (seed, n, a.shape, o_a.shape)
            assert np.array_equal(a, o_a), (seed, n)
            assert np.array_equal(s, o_s), (seed, n, s, o_s)
            key = 'ae_%d_%d' % (seed, n)
            out[key + '_ans'] = a
            out[key + '_scores'] = s
            if n == 0:
                out[key + '_val_k'] = tk['val_k'][0]
                out[key + '_loc_k'] = tk['loc_k'][0]
                out[key + '_tag_k'] = tk['tag_k'][0]
            scenes.append((seed, n, a.shape[0]))
        out['ae_%d_meta' % seed] = np.array([J, R, T, len(people)] + list(people))
    print('AE scenes (seed, n, persons):', scenes)

    np.savez_compressed(os.path.join(HERE, 'golden.npz'), **out)
    print('wrote', os.path.join(HERE, 'golden.npz'),
          os.path.getsize(os.path.join(HERE, 'golden.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
