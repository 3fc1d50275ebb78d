#!/usr/bin/env python
"""P3 of the parity protocol: the GPU pipeline end to end against the FULL CPU oracle pipeline
(network, flip-TTA merge, parser all on the CPU) on the bench's synthetic scenes.  The reference parser is a
chaotic function of its inputs (a 1e-7 heatmap difference can flip an argmax or a Hungarian assignment,
SURVEY.md section 7), so this is REPORTED, not asserted: per-image identical-record rate, per-joint agreement,
and for every disagreeing joint the decision margin on the CPU maps (how much lower the CPU heatmap is at the
GPU's pick than at its own pick).
    python tools/p3_agreement.py --images 64 > profiles/r02_p3_agreement.txt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from litepose_amd import arch_zoo, config, engine  # noqa: E402
from oracle import group_ref, inference_ref, net_ref, oks, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=64)
ap.add_argument('--head-gain', type=float, default=0.25)
ap.add_argument('--arch', default='search-XS')
ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'],
                help='bf16: the agreement of the bf16-storage network with the fp32 CPU pipeline (a budget, not parity)')
a = ap.parse_args()
arch = arch_zoo.get(a.arch)
R = arch['img_size']
cfg = config.apply_arch(config.get_cfg(), arch)
sd = synth.make_state_dict(arch, seed=1234, head_gain=a.head_gain)
N = a.images
eng = engine.PoseEngine(cfg, arch, sd, person_capacity=64, storage=a.storage, options=engine.options_from_env())
x = synth.make_images(N, R, seed=100)
off0, off1 = synth.lowres_offsets(200, N, 14, R)
f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
ans, count, scores = [t.cpu().numpy() for t in eng.infer_batch(x.cuda(), offsets=offs)]
gdet, gtag = [t.cpu().numpy() for t in eng.last_maps()]
torch.set_num_threads(min(os.cpu_count() or 1, 64))
with torch.no_grad():
    o = net_ref.forward(x, sd, arch)
    of = net_ref.forward(torch.flip(x, [3]), sd, arch)
    o = [o[0] + torch.from_numpy(off0), o[1] + torch.from_numpy(off1)]
    of = [of[0] + torch.from_numpy(f0), of[1] + torch.from_numpy(f1)]
    fh, tg = inference_ref.merge(o, of, inference_ref.TestCfg(), (R, R))
fh, tg = fh.numpy(), tg.numpy()
ora = group_ref.HeatmapParser(group_ref.Params())
print('# P3: GPU pipeline (%s storage) vs full fp32 CPU oracle pipeline, LitePose-%s@%d, %d synthetic scenes (bench inputs), '
      'head_gain %.2f' % (a.storage, a.arch.split('-')[-1], R, N, a.head_gain))
print('heatmap max-abs diff GPU vs CPU maps: det %.3e  tag %.3e' % (float(np.abs(gdet - fh).max()),
                                                                    float(np.abs(gtag - tg).max())))
same_img = same_cnt = same_kp = 0
joints = agree = near = 0
margins = []
oks_vals = []
for n in range(N):
    a_cpu, s_cpu = ora.parse_image(fh[n], tg[n])
    k = int(count[n])
    a_gpu = ans[n, :k]
    oks_vals += oks.image_oks([a_cpu[q] for q in range(a_cpu.shape[0])], [a_gpu[q] for q in range(min(k, ans.shape[1]))])
    if k == a_cpu.shape[0]:
        same_cnt += 1
    if k == a_cpu.shape[0] and np.array_equal(a_gpu, a_cpu):
        same_img += 1
    if k == a_cpu.shape[0] and np.array_equal(a_gpu[..., :2], a_cpu[..., :2]) and \
            np.array_equal(a_gpu[..., 2] > 0, a_cpu[..., 2] > 0):
        same_kp += 1
    for p in range(min(k, a_cpu.shape[0])):
        for j in range(14):
            joints += 1
            g, c = a_gpu[p, j], a_cpu[p, j]
            if (g[2] > 0) == (c[2] > 0) and (c[2] <= 0 or max(abs(g[0] - c[0]), abs(g[1] - c[1])) <= 1.0):
                near += 1
            if np.array_equal(g[:2], c[:2]) and (g[2] > 0) == (c[2] > 0):
                agree += 1
            elif c[2] > 0 and g[2] > 0:
                yc, xc, yg, xg = int(c[1]), int(c[0]), int(g[1]), int(g[0])
                margins.append(float(fh[n, j, yc, xc] - fh[n, j, yg, xg]))
            else:
                margins.append(float('nan'))
print('images with the same person count: %d / %d' % (same_cnt, N))
print('images with identical keypoints (persons, order, joint presence, x/y incl. quarter offsets): %d / %d'
      % (same_kp, N))
print('images whose records are also bit-identical in the float columns (heatmap value, tags; these inherit the '
      '1e-7 heatmap difference): %d / %d' % (same_img, N))
print('joints compared (persons matched by order): %d, identical position+presence: %d (%.4f %%)'
      % (joints, agree, 100.0 * agree / max(1, joints)))
print('joints with the same presence and a position within 1 px (the granularity OKS / mAP sees): %d (%.4f %%)'
      % (near, 100.0 * near / max(1, joints)))
so = oks.summary(oks_vals)
print('OKS of the device records against the persons of the CPU pipeline (greedy one-to-one match per image, CrowdPose sigmas, '
      'area = keypoint box; oracle/oks.py): %d persons, mean %s, 5th percentile %s, minimum %s'
      % (so['persons'], so['mean'], so['p05'], so['min']))
m = np.asarray(margins, np.float64)
print('disagreeing joints: %d (presence flips: %d)' % (len(m), int(np.isnan(m).sum())))
m = m[~np.isnan(m)]
if len(m):
    edges = [-1.0, 0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1.0]
    hist, _ = np.histogram(m, bins=edges)
    print('decision margin histogram (CPU heatmap at its own pick minus at the GPU pick):')
    for lo, hi, cnt in zip(edges[:-1], edges[1:], hist):
        print('  [%8.1e, %8.1e): %d' % (lo, hi, cnt))
