"""Drop-in for ``core.inference`` (reference lib/core/inference.py:75-208).

Same call signatures and return shapes as the reference so the ``valid.py`` loop body
runs unchanged; the arithmetic (stage-0 upsample, stage/flip averaging, FLIP_CONFIG
permutation, projection) is one native call, ``lp_tta_merge``.
"""
import ctypes as C

import torch

from .. import _native as nv

FLIP_CONFIG = {      # lib/dataset/transforms/build.py:15-28 (a constant table of the datasets)
    'COCO': [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15],
    'COCO_WITH_CENTER': [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 17],
    'CROWDPOSE': [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 12, 13],
    'CROWDPOSE_WITH_CENTER': [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 12, 13, 14],
}


def flip_index_for(cfg):
    if 'coco' in cfg.DATASET.DATASET:
        name = 'COCO'
    elif 'crowd_pose' in cfg.DATASET.DATASET:
        name = 'CROWDPOSE'
    else:
        raise ValueError('Please implement flip_index for new dataset: %s.' % cfg.DATASET.DATASET)
    return FLIP_CONFIG[name + '_WITH_CENTER'] if cfg.DATASET.WITH_CENTER else FLIP_CONFIG[name]


def _check_cfg(cfg):
    if list(cfg.LOSS.WITH_HEATMAPS_LOSS) != [True, True] or list(cfg.LOSS.WITH_AE_LOSS) != [True, False] \
            or list(cfg.TEST.WITH_HEATMAPS) != [True, True] or list(cfg.TEST.WITH_AE) != [True, False] \
            or not cfg.MODEL.TAG_PER_JOINT:
        raise NotImplementedError('the accelerated path implements the LitePose (mobile.yaml) stage layout')


def used_joints(cfg):
    """Joints the merged maps carry: DATASET.NUM_JOINTS (which already counts the centre joint when
    DATASET.WITH_CENTER, default.py:175) minus the centre when TEST.IGNORE_CENTER (inference.py:148-150)."""
    jn = int(cfg.DATASET.NUM_JOINTS)
    return jn - 1 if (cfg.DATASET.WITH_CENTER and cfg.TEST.IGNORE_CENTER) else jn


_ws_cache = {}


def tta_merge(cfg, outs, outs_flip, size_projected, det=None, tag=None, ws=None):
    """outs/outs_flip: [out0, out1] device tensors (outs_flip may be None).
    Returns (final_heatmaps [N,J,Hp,Wp], tags [N,J,Hp,Wp,T]).
    ``ws``: caller-owned scratch (uint8, >= lp_tta_workspace_bytes) -- the engine passes one per
    lane/stream.  Without it a scratch buffer private to (device, current stream) is used: the stage-1
    intermediate lives there between the two kernels, so two streams must never share it."""
    _check_cfg(cfg)
    lib = nv.lib()
    out0, out1 = outs
    Jn = int(cfg.DATASET.NUM_JOINTS)            # joints per network stage (incl. the centre joint, if any)
    J = used_joints(cfg)
    N, C0, h0, w0 = out0.shape
    _, C1, h1, w1 = out1.shape
    if C0 != 2 * Jn or C1 != Jn:
        raise ValueError('unexpected head channels')
    if size_projected:
        Wp, Hp = int(size_projected[0]), int(size_projected[1])
    else:
        Wp, Hp = w1, h1
    T = 2 if outs_flip is not None else 1
    dev = out0.device
    if det is None:
        det = torch.empty((N, J, Hp, Wp), dtype=torch.float32, device=dev)
    if tag is None:
        tag = torch.empty((N, J, Hp, Wp, T), dtype=torch.float32, device=dev)
    need = int(lib.lp_tta_workspace_bytes(N, J, h1, w1))
    if ws is None:
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < need:
            # the old buffer may still be read by kernels queued on this stream: the caching allocator
            # only re-uses it for allocations made on the same stream, i.e. after them
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            _ws_cache[key] = ws
    elif ws.numel() < need or not ws.is_cuda:
        raise ValueError('tta_merge: workspace too small (%d < %d bytes)' % (ws.numel(), need))
    # the flip permutation is applied before the centre joint is dropped; it maps the centre to itself
    fi = (C.c_int32 * J)(*flip_index_for(cfg)[:J])
    o0f = nv.dptr(outs_flip[0]) if outs_flip is not None else None
    o1f = nv.dptr(outs_flip[1]) if outs_flip is not None else None
    nv.check(lib.lp_tta_merge_ex(nv.dptr(out0), nv.dptr(out1), o0f, o1f, N, J, C0, C1, Jn, h0, w0, h1, w1, Hp, Wp,
                                 C.cast(fi, C.c_void_p), nv.dptr(det), nv.dptr(tag), nv.dptr(ws), need,
                                 nv.stream_ptr()), 'lp_tta_merge_ex')
    return det, tag


def stage_add_supported(N, J, h0, w0, h1, w1):
    """The gate of ``lp_tta_stage_add`` (additive maps ride on the exact x2 stage merge only)."""
    import os
    return (h1 == 2 * h0 and w1 == 2 * w0 and w1 % 32 == 0
            and h1 % 8 == 0 and N * J <= 65535)


def tta_stage(cfg, outs, outs_flip, mid, add=None):
    """First half of ``tta_merge`` only: stage merge at the stage-1 resolution (inference.py:84-146) into the
    caller's ``mid`` buffer (uint8, >= lp_tta_workspace_bytes) laid out [N][4][J][h1][w1] = heat, heat_flip,
    tag, tag_flip.  ``lp_parse_mid`` / ``tta_project`` consume it.  Returns (N, J, h1, w1, T).
    ``add``: optional (add0, add1) of the shapes [N (+N mirrored), C, h, w] of the outputs, added as the outputs are
    read (``lp_tta_stage_add``; check ``stage_add_supported`` first)."""
    _check_cfg(cfg)
    lib = nv.lib()
    out0, out1 = outs
    Jn = int(cfg.DATASET.NUM_JOINTS)
    J = used_joints(cfg)
    N, C0, h0, w0 = out0.shape
    _, C1, h1, w1 = out1.shape
    if C0 != 2 * Jn or C1 != Jn:
        raise ValueError('unexpected head channels')
    T = 2 if outs_flip is not None else 1
    need = int(lib.lp_tta_workspace_bytes(N, J, h1, w1))
    if mid.numel() < need or not mid.is_cuda:
        raise ValueError('tta_stage: mid buffer too small (%d < %d bytes)' % (mid.numel(), need))
    fi = (C.c_int32 * J)(*flip_index_for(cfg)[:J])
    o0f = nv.dptr(outs_flip[0]) if outs_flip is not None else None
    o1f = nv.dptr(outs_flip[1]) if outs_flip is not None else None
    if add is not None:
        a0, a1 = add
        nf = 2 * N if outs_flip is not None else N
        if tuple(a0.shape) != (nf, C0, h0, w0) or tuple(a1.shape) != (nf, C1, h1, w1):
            raise ValueError('additive maps must have the shapes of the stacked outputs')
        if a0.dtype != torch.float32 or a1.dtype != torch.float32:
            raise ValueError('additive maps must be float32 (the kernel reads them as fp32 arrays)')
        a0f = nv.dptr(a0[N:]) if outs_flip is not None else None
        a1f = nv.dptr(a1[N:]) if outs_flip is not None else None
        nv.check(lib.lp_tta_stage_add(nv.dptr(out0), nv.dptr(out1), o0f, o1f, nv.dptr(a0[:N]), nv.dptr(a1[:N]), a0f,
                                      a1f, N, J, C0, C1, Jn, h0, w0, h1, w1, C.cast(fi, C.c_void_p), nv.dptr(mid),
                                      mid.numel(), nv.stream_ptr()), 'lp_tta_stage_add')
        return N, J, h1, w1, T
    nv.check(lib.lp_tta_stage(nv.dptr(out0), nv.dptr(out1), o0f, o1f, N, J, C0, C1, Jn, h0, w0, h1, w1,
                              C.cast(fi, C.c_void_p), nv.dptr(mid), mid.numel(), nv.stream_ptr()), 'lp_tta_stage')
    return N, J, h1, w1, T


def tta_project(mid, N, J, h1, w1, size_projected, T, det=None, tag=None, det_only=False):
    """Second half of ``tta_merge``: projection of ``mid`` to ``size_projected`` (W, H) + flip average.
    ``det_only``: heatmaps only (exact x2 projection; ``lp_parse_dm`` evaluates the tags from ``mid``)."""
    Wp, Hp = int(size_projected[0]), int(size_projected[1])
    if det is None:
        det = torch.empty((N, J, Hp, Wp), dtype=torch.float32, device=mid.device)
    if tag is None and not det_only:
        tag = torch.empty((N, J, Hp, Wp, T), dtype=torch.float32, device=mid.device)
    nv.check(nv.lib().lp_tta_project(nv.dptr(mid), N, J, h1, w1, Hp, Wp, T, nv.dptr(det),
                                     None if det_only else nv.dptr(tag), nv.stream_ptr()), 'lp_tta_project')
    return det, (None if det_only else tag)


class _Merged(list):
    """What get_multi_stage_outputs hands to aggregate_results: the reference passes two
    lists of per-flip maps; here the merge has already been done natively."""
    pass


def get_multi_stage_outputs(cfg, model, image, with_flip=False, project2image=False, size_projected=None):
    """inference.py:75-173.  Returns (outputs, heatmaps, tags) like the reference; the
    heatmaps/tags lists carry the natively merged result for ``aggregate_results``."""
    if with_flip and hasattr(model, 'forward_native'):
        both = model.forward_native(image, flip=2)
        n = image.shape[0]
        outs = [both[0][:n], both[1][:n]]
        outs_f = [both[0][n:], both[1][n:]]
    else:
        outs = model(image)
        outs_f = model(torch.flip(image, [3])) if with_flip else None
    sp = size_projected if (project2image and size_projected) else None
    det, tag = tta_merge(cfg, outs, outs_f, sp)
    heatmaps = _Merged([det])
    tags = _Merged([tag])
    outputs = list(outs) + (list(outs_f) if outs_f is not None else [])
    return outputs, heatmaps, tags


def resize_maps(det, tag, size_hw):
    """Bilinear resize (``interpolate(..., align_corners=False)``) of merged maps det [N,J,h,w] / tag [N,J,h,w,T] to
    ``size_hw`` = (H, W) with the projection kernel of ``lp_tta_project``: the merged maps are laid out as its ``mid`` input
    (heat = heat_flip = det, so the flip average (P(det) + P(det)) / 2 is P(det) exactly; tag planes de-interleaved).
    Used by ``aggregate_results`` for TEST.PROJECT2IMAGE = False (inference.py:180-189, 201-206)."""
    N, J, h, w = det.shape
    T = tag.shape[4] if tag is not None else 1
    if T > 2:
        raise NotImplementedError('at most two tag maps per joint (flip test)')
    mid = torch.empty((N, 4, J, h, w), dtype=torch.float32, device=det.device)
    mid[:, 0] = det
    mid[:, 1] = det
    if tag is not None:
        mid[:, 2] = tag[..., 0]
        mid[:, 3] = tag[..., T - 1]
    else:
        mid[:, 2:] = 0
    return tta_project(mid, N, J, h, w, (int(size_hw[1]), int(size_hw[0])), T)


def aggregate_results(cfg, scale_factor, final_heatmaps, tags_list, heatmaps, tags):
    """inference.py:176-208.  Called once per TEST.SCALE_FACTOR entry (valid.py:207-222):
    tags are kept from scale 1 only (or from the single scale), heatmaps of every scale --
    already flip-averaged (and, with TEST.PROJECT2IMAGE, projected to the common base size) by the native merge -- are
    summed with ``lp_maps_accumulate``; the caller divides by len(SCALE_FACTOR) (valid.py:224).
    TEST.PROJECT2IMAGE = False (round 6): the maps of a scale come at that scale's stage-1 resolution; the tags of scale 1
    (:180-189) and the flip-averaged heatmaps of every later scale (:201-206) are resized to the first scale's maps."""
    if not isinstance(heatmaps, _Merged) or not isinstance(tags, _Merged):
        raise TypeError('heatmaps/tags must come from litepose_amd.core.inference.get_multi_stage_outputs')
    det, tag = heatmaps[0], tags[0]
    p2i = bool(cfg.TEST.PROJECT2IMAGE)
    resized = None
    if final_heatmaps is not None and not p2i and tuple(det.shape[2:4]) != tuple(final_heatmaps.shape[2:4]):
        resized = resize_maps(det, tag, final_heatmaps.shape[2:4])
    if scale_factor == 1 or len(cfg.TEST.SCALE_FACTOR) == 1:
        if final_heatmaps is not None and tuple(tag.shape[2:4]) != tuple(final_heatmaps.shape[2:4]):
            if p2i:
                raise ValueError('TEST.PROJECT2IMAGE: maps of different scales must share the projected size')
            tag = resized[1]                    # inference.py:180-189
        tags_list.append(tag)       # already [N,J,H,W,T]; torch.cat(tags_list, dim=4) is then a no-op
    if final_heatmaps is None:
        return det, tags_list
    if resized is not None:
        det = resized[0]                        # inference.py:201-206
    if tuple(det.shape) != tuple(final_heatmaps.shape):
        raise ValueError('TEST.PROJECT2IMAGE: maps of different scales must share the projected size')
    if not final_heatmaps.is_contiguous() or final_heatmaps.dtype != torch.float32:
        raise ValueError('final_heatmaps must be a contiguous float32 device tensor')
    nv.check(nv.lib().lp_maps_accumulate(nv.dptr(final_heatmaps), nv.dptr(det), det.numel(), nv.stream_ptr()),
             'lp_maps_accumulate')
    return final_heatmaps, tags_list
