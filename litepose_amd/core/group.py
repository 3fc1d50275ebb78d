"""Drop-in for ``core.group`` (reference lib/core/group.py:100-291).

``HeatmapParser(cfg).parse(det, tag, adjust, refine) -> (ans, scores)`` with the
reference's return shapes (``ans = [ndarray[P, J, 3+T]]`` for image 0, ``scores`` a list
of P floats), computed by the native kernels (lp_peaks_topk / lp_group /
lp_adjust_refine).  ``parse_batch`` is the batched generalisation: per image the
result equals the reference's batch-1 result.
"""
import ctypes as C

import numpy as np
import torch

from .. import _native as nv


class Params(object):
    # group.py:100-120
    def __init__(self, cfg):
        self.num_joints = cfg.DATASET.NUM_JOINTS
        self.max_num_people = cfg.DATASET.MAX_NUM_PEOPLE
        self.detection_threshold = cfg.TEST.DETECTION_THRESHOLD
        self.tag_threshold = cfg.TEST.TAG_THRESHOLD
        self.use_detection_val = cfg.TEST.USE_DETECTION_VAL
        self.ignore_too_much = cfg.TEST.IGNORE_TOO_MUCH
        if cfg.DATASET.WITH_CENTER and cfg.TEST.IGNORE_CENTER:
            self.num_joints -= 1
        if cfg.DATASET.WITH_CENTER and not cfg.TEST.IGNORE_CENTER:
            self.joint_order = [i - 1 for i in
                                [18, 1, 2, 3, 4, 5, 6, 7, 12, 13, 8, 9, 10, 11, 14, 15, 16, 17]]
        else:
            self.joint_order = [i - 1 for i in
                                [1, 2, 3, 4, 5, 6, 7, 12, 13, 8, 9, 10, 11, 14, 15, 16, 17]]


class HeatmapParser(object):
    def __init__(self, cfg, person_capacity=None):
        self.params = Params(cfg)
        self.tag_per_joint = cfg.MODEL.TAG_PER_JOINT
        self.nms_kernel = int(cfg.TEST.NMS_KERNEL)
        if int(cfg.TEST.NMS_PADDING) * 2 + 1 != self.nms_kernel:
            raise ValueError('NMS_PADDING must be NMS_KERNEL // 2')
        p = self.params
        if p.detection_threshold < 0:
            raise ValueError('DETECTION_THRESHOLD must be >= 0')
        # every joint can open at most max_num_people new persons (group.py:90-94)
        self.person_capacity = int(person_capacity or min(1024, p.num_joints * p.max_num_people))
        self._lib = nv.lib()
        q = nv.LpParseParams()
        q.num_joints = p.num_joints
        q.max_num_people = p.max_num_people
        q.detection_threshold = float(p.detection_threshold)
        q.tag_threshold = float(p.tag_threshold)
        q.use_detection_val = int(bool(p.use_detection_val))
        q.ignore_too_much = int(bool(p.ignore_too_much))
        q.nms_kernel = self.nms_kernel
        q.tag_per_joint = int(bool(self.tag_per_joint))
        for i in range(p.num_joints):
            q.joint_order[i] = int(p.joint_order[i])
        self._q = q
        self._ws = {}

    # ---- helpers ----------------------------------------------------------------
    @staticmethod
    def _dev(t):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.ascontiguousarray(t))
        if not t.is_cuda:
            t = t.cuda()
        return t.to(torch.float32).contiguous()

    def _scratch(self, name, nbytes, device):
        t = self._ws.get((name, device))
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self._ws[(name, device)] = t
        return t

    # ---- reference-shaped pieces ----------------------------------------------------
    def top_k(self, det, tag):
        """group.py:141-176.  Returns the reference's dict of NumPy arrays."""
        det, tag = self._dev(det), self._dev(tag)
        N, J, H, W = det.shape
        T = tag.shape[4]
        M = self.params.max_num_people
        val_k = torch.empty((N, J, M), dtype=torch.float32, device=det.device)
        ind_k = torch.empty((N, J, M), dtype=torch.int32, device=det.device)
        tag_k = torch.empty((N, J, M, T), dtype=torch.float32, device=det.device)
        nv.check(self._lib.lp_peaks_topk(nv.dptr(det), nv.dptr(tag), N, J, H, W, T, C.byref(self._q),
                                         nv.dptr(val_k), nv.dptr(ind_k), nv.dptr(tag_k), nv.stream_ptr()),
                 'lp_peaks_topk')
        ind = ind_k.cpu().numpy().astype(np.int64)
        return {'tag_k': tag_k.cpu().numpy(), 'loc_k': np.stack((ind % W, ind // W), axis=3),
                'val_k': val_k.cpu().numpy()}

    def parse_batch_device(self, det, tag, adjust=True, refine=True):
        """Device in / device out: (ans [N,pcap,J,3+T], count [N] i32, scores [N,pcap])."""
        det, tag = self._dev(det), self._dev(tag)
        N, J, H, W = det.shape
        T = tag.shape[4]
        M = self.params.max_num_people
        pcap = self.person_capacity
        dev = det.device
        ans = torch.empty((N, pcap, J, 3 + T), dtype=torch.float32, device=dev)
        count = torch.empty((N,), dtype=torch.int32, device=dev)
        scores = torch.empty((N, pcap), dtype=torch.float32, device=dev)
        need = int(self._lib.lp_parse_workspace_bytes(N, J, M, T, pcap))
        ws = self._scratch('parse', need, dev)
        nv.check(self._lib.lp_parse(nv.dptr(det), nv.dptr(tag), N, J, H, W, T, C.byref(self._q), pcap,
                                    int(bool(adjust)), int(bool(refine)), nv.dptr(ans), nv.dptr(count),
                                    nv.dptr(scores), nv.dptr(ws), need, nv.stream_ptr()), 'lp_parse')
        return ans, count, scores

    def parse_batch(self, det, tag, adjust=True, refine=True):
        """Per image: (ans ndarray[P,J,3+T] float32, scores ndarray[P] float32)."""
        ans, count, scores = self.parse_batch_device(det, tag, adjust, refine)
        cnt = count.cpu().numpy()
        a = ans.cpu().numpy()
        s = scores.cpu().numpy()
        out = []
        for n in range(a.shape[0]):
            if cnt[n] < 0:
                raise nv.LitePoseNativeError('grouping failed on image %d (solver guard tripped)' % n)
            if cnt[n] > self.person_capacity:
                raise nv.LitePoseNativeError('person capacity %d exceeded (%d)' % (self.person_capacity, cnt[n]))
            out.append((a[n, :cnt[n]].copy(), s[n, :cnt[n]].copy()))
        return out

    def parse(self, det, tag, adjust=True, refine=True):
        """group.py:269-291: results of image 0 (the reference only supports batch 1)."""
        a, s = self.parse_batch(det[:1], tag[:1], adjust, refine)[0]
        return [a], [x for x in s]
