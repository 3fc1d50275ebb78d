#!/usr/bin/env python
"""Golden fixture for the result writer: runs the REAL reference
`CrowdPoseDataset.evaluate` -> `_write_coco_keypoint_results` (lib/dataset/CrowdPoseDataset.py:147-279)
on seeded synthetic predictions and stores the JSON it writes (tests/golden/results_golden.json)
next to the inputs (results_inputs.npz).  Build container only (needs /root/reference).

The dataset class cannot be constructed here (cv2 / json_tricks / crowdposetools are absent), so
those imports are stubbed, the instance is created without __init__, and the COCO evaluation step
(`_do_python_keypoint_eval`) is replaced by a constant -- only the formatting code runs.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'


def main():
    np.float = float                                 # removed from NumPy 2; the reference still uses it
    for name in ('cv2', 'crowdposetools', 'crowdposetools.cocoeval', 'utils', 'utils.zipreader'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['crowdposetools.cocoeval'].COCOeval = object
    sys.modules['utils'].zipreader = sys.modules['utils.zipreader']
    sys.modules['json_tricks'] = json
    spec = importlib.util.spec_from_file_location('ref_crowdpose_ds', os.path.join(REF, 'lib/dataset/CrowdPoseDataset.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ds = object.__new__(mod.CrowdPoseDataset)
    rng = np.random.default_rng(77)
    n_img, J = 4, 14
    ids = [100103, 100104, 100200, 100317]
    ds.ids = ids
    ds.dataset = 'test'
    ds.classes = ['__background__', 'person']
    ds._class_to_coco_ind = {'person': 1}
    ds.coco = types.SimpleNamespace(loadImgs=lambda i: [{'file_name': '%d.jpg' % i}])
    ds._do_python_keypoint_eval = lambda res_file, res_folder: [('AP', 0.0)]
    counts = [3, 0, 1, 5]
    preds, scores = [], []
    for n in range(n_img):
        persons = []
        for p in range(counts[n]):
            k = np.zeros((J, 5), np.float32)
            k[:, 0:2] = rng.uniform(0, 256, size=(J, 2)).astype(np.float32)
            k[:, 2] = rng.uniform(0, 1, size=J).astype(np.float32)
            k[:, 3:5] = rng.normal(size=(J, 2)).astype(np.float32)
            k[rng.random(J) < 0.2] = 0                  # missing joints stay all-zero rows
            persons.append(k)
        preds.append(persons)
        scores.append([float(np.float32(np.mean(k[:, 2]))) for k in persons])
    NS = types.SimpleNamespace
    cfg = NS(DATASET=NS(WITH_CENTER=False), TEST=NS(IGNORE_CENTER=True))
    with tempfile.TemporaryDirectory() as d:
        ds.evaluate(cfg, preds, scores, d)
        txt = open(os.path.join(d, 'results', 'keypoints_test_results.json')).read()
    open(os.path.join(HERE, 'results_golden.json'), 'w').write(txt)
    flat = np.zeros((n_img, max(counts), J, 5), np.float32)
    sc = np.zeros((n_img, max(counts)), np.float32)
    for n in range(n_img):
        for p in range(counts[n]):
            flat[n, p] = preds[n][p]
            sc[n, p] = scores[n][p]
    np.savez_compressed(os.path.join(HERE, 'results_inputs.npz'), kpts=flat, count=np.array(counts, np.int32),
                        scores=sc, ids=np.array(ids, np.int64))
    print('wrote results_golden.json (%d entries)' % len(json.loads(txt)))


if __name__ == '__main__':
    main()
