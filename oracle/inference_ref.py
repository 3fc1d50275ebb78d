"""Flip-TTA / multi-stage merge / projection oracle.  TEST INFRASTRUCTURE.

torch fp32 restatement (floating-point path) of
  * /root/reference/lib/core/inference.py:75-173  get_multi_stage_outputs
  * /root/reference/lib/core/inference.py:176-208 aggregate_results
  * /root/reference/valid.py:224-225              /len(SCALE_FACTOR), cat(dim=4)
  * /root/reference/lib/dataset/transforms/build.py:15-28  FLIP_CONFIG
for the single-scale case (TEST.SCALE_FACTOR == [1], mobile.yaml:66) and, in
``merge_multiscale``, the multi-scale loop of valid.py:207-225.
"""
import torch
import torch.nn.functional as F

FLIP_CONFIG = {
    'COCO': [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15],
    'COCO_WITH_CENTER': [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 17],
    'CROWDPOSE': [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 12, 13],
    'CROWDPOSE_WITH_CENTER': [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 12, 13, 14],
}


class TestCfg(object):
    """The TEST/LOSS/DATASET fields the merge reads (mobile.yaml:37-72)."""

    def __init__(self, num_joints=14, dataset='crowd_pose_kpt',
                 with_heatmaps_loss=(True, True), with_ae_loss=(True, False),
                 with_heatmaps=(True, True), with_ae=(True, False),
                 tag_per_joint=True, flip_test=True, project2image=True, with_center=False,
                 ignore_center=True):
        self.num_joints = num_joints
        self.dataset = dataset
        self.with_heatmaps_loss = with_heatmaps_loss
        self.with_ae_loss = with_ae_loss
        self.with_heatmaps = with_heatmaps
        self.with_ae = with_ae
        self.tag_per_joint = tag_per_joint
        self.flip_test = flip_test
        self.project2image = project2image
        self.with_center = with_center          # num_joints then counts the centre joint (default.py:175)
        self.ignore_center = ignore_center

    def flip_index(self):
        sfx = '_WITH_CENTER' if self.with_center else ''
        if 'coco' in self.dataset:
            return FLIP_CONFIG['COCO' + sfx]
        if 'crowd_pose' in self.dataset:
            return FLIP_CONFIG['CROWDPOSE' + sfx]
        raise ValueError(self.dataset)


def _up(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


def _one_pass(outputs, tc, flip):
    J = tc.num_joints
    heatmaps_avg = 0
    num_heatmaps = 0
    tags = []
    fi = tc.flip_index()
    for i, output in enumerate(outputs):
        if len(outputs) > 1 and i != len(outputs) - 1:
            output = _up(output, (outputs[-1].size(2), outputs[-1].size(3)))
        if flip:
            output = torch.flip(output, [3])
        offset_feat = J if tc.with_heatmaps_loss[i] else 0
        if tc.with_heatmaps_loss[i] and tc.with_heatmaps[i]:
            h = output[:, :J]
            if flip:
                h = h[:, fi, :, :]
            heatmaps_avg = heatmaps_avg + h          # inference.py:99 / :135 (0 + a + b)
            num_heatmaps += 1
        if tc.with_ae_loss[i] and tc.with_ae[i]:
            t = output[:, offset_feat:]
            if flip and tc.tag_per_joint:
                t = t[:, fi, :, :]
            tags.append(t)
    return heatmaps_avg / num_heatmaps, tags


def merge(outputs, outputs_flip, tc, size_projected):
    """outputs / outputs_flip: network outputs for image and flip(image,[3]).

    Returns (final_heatmaps [N,J,H,W], tags [N,J,H,W,T]) exactly as
    ``valid.py`` hands them to ``HeatmapParser.parse``."""
    heatmaps = []
    tags = []
    h, t = _one_pass(outputs, tc, False)
    heatmaps.append(h)
    tags += t
    if tc.flip_test:
        h, t = _one_pass(outputs_flip, tc, True)
        heatmaps.append(h)
        tags += t
    if tc.with_center and tc.ignore_center:            # inference.py:148-150
        heatmaps = [hms[:, :-1] for hms in heatmaps]
        tags = [tms[:, :-1] for tms in tags]
    if tc.project2image and size_projected:
        size = (size_projected[1], size_projected[0])
        heatmaps = [_up(hms, size) for hms in heatmaps]
        tags = [_up(tms, size) for tms in tags]
    # aggregate_results, scale_factor == 1, final_heatmaps is None
    tags_list = [torch.unsqueeze(tms, dim=4) for tms in tags]
    final = (heatmaps[0] + heatmaps[1]) / 2.0 if tc.flip_test else heatmaps[0]
    final = final / 1.0                                # valid.py:224 (one scale)
    return final, torch.cat(tags_list, dim=4)


def run(model_fn, image, tc, size_projected=None):
    outs = model_fn(image)
    outs_f = model_fn(torch.flip(image, [3])) if tc.flip_test else None
    if size_projected is None:
        size_projected = (image.shape[3], image.shape[2])
    return merge(outs, outs_f, tc, size_projected)


def project_pass(outputs, outputs_flip, tc, size_projected):
    """get_multi_stage_outputs (inference.py:75-173) on precomputed network outputs:
    returns the per-flip lists (heatmaps, tags) the reference hands to aggregate_results."""
    heatmaps = []
    tags = []
    h, t = _one_pass(outputs, tc, False)
    heatmaps.append(h)
    tags += t
    if tc.flip_test:
        h, t = _one_pass(outputs_flip, tc, True)
        heatmaps.append(h)
        tags += t
    if tc.with_center and tc.ignore_center:            # inference.py:148-150
        heatmaps = [hms[:, :-1] for hms in heatmaps]
        tags = [tms[:, :-1] for tms in tags]
    if tc.project2image and size_projected:
        size = (size_projected[1], size_projected[0])
        heatmaps = [_up(hms, size) for hms in heatmaps]
        tags = [_up(tms, size) for tms in tags]
    return heatmaps, tags


def merge_multiscale(per_scale, tc, base_size):
    """valid.py:207-225 with aggregate_results (inference.py:176-208) inlined.

    per_scale: list of (scale_factor, outputs, outputs_flip); visited in descending scale
    order like ``sorted(cfg.TEST.SCALE_FACTOR, reverse=True)``.  Tags are taken from scale 1
    only (inference.py:179); heatmaps are summed in visiting order, then divided by the
    number of scales."""
    per_scale = sorted(per_scale, key=lambda e: e[0], reverse=True)
    n_scales = len(per_scale)
    final = None
    tags_list = []
    for s, outs, outs_f in per_scale:
        heatmaps, tags = project_pass(outs, outs_f, tc, base_size)
        if s == 1 or n_scales == 1:
            if final is not None and not tc.project2image:
                tags = [_up(t, (final.size(2), final.size(3))) for t in tags]
            for t in tags:
                tags_list.append(torch.unsqueeze(t, dim=4))
        avg = (heatmaps[0] + heatmaps[1]) / 2.0 if tc.flip_test else heatmaps[0]
        if final is None:
            final = avg
        elif tc.project2image:
            final = final + avg
        else:
            final = final + _up(avg, (final.size(2), final.size(3)))
    final = final / float(n_scales)
    return final, torch.cat(tags_list, dim=4)
