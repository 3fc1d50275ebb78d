"""Drop-in for ``models.pose_mobilenet`` (reference lib/models/pose_mobilenet.py).

``get_pose_net(cfg, is_train, cfg_arch)`` returns a ``LitePose`` object with the
subset of the ``nn.Module`` surface valid.py touches (valid.py:125-165,194):
``__call__/forward``, ``eval()``, ``cuda()``, ``load_state_dict(sd, strict)``,
``state_dict()``.  The network itself is the native engine behind the C ABI
(lp_net_*): Python only moves pointers.
"""
import ctypes as C
from collections import OrderedDict

import torch

from .. import _native as nv


def _arch_struct(cfg, cfg_arch):
    a = nv.LpArch()
    a.input_channel = int(cfg_arch['input_channel'])
    bs = cfg_arch['backbone_setting']
    if len(bs) > nv.LP_MAX_STAGES:
        raise ValueError('too many stages')
    a.num_stages = len(bs)
    for s, st in enumerate(bs):
        a.num_blocks[s] = int(st['num_blocks'])
        a.stride[s] = int(st['stride'])
        a.channel[s] = int(st['channel'])
        for b in range(st['num_blocks']):
            t, k = st['block_setting'][b]
            a.expand[s][b] = int(t)
            a.kernel[s][b] = int(k)
    extra = cfg.MODEL.EXTRA
    a.num_deconv = int(extra.NUM_DECONV_LAYERS)
    if any(int(k) != 4 for k in extra.NUM_DECONV_KERNELS[:a.num_deconv]):
        raise ValueError('only NUM_DECONV_KERNELS == 4 is supported on this path')
    for i, f in enumerate(cfg_arch['deconv_setting'][:a.num_deconv]):
        a.deconv_filters[i] = int(f)
    dim_tag = cfg.MODEL.NUM_JOINTS if cfg.MODEL.TAG_PER_JOINT else 1
    for i in range(1, a.num_deconv):        # pose_mobilenet.py:92-98
        oup = (cfg.MODEL.NUM_JOINTS if cfg.LOSS.WITH_HEATMAPS_LOSS[i - 1] else 0) + \
              (dim_tag if cfg.LOSS.WITH_AE_LOSS[i - 1] else 0)
        a.head_channels[i - 1] = int(oup)
    return a


STORAGE = {'f32': 0, 'fp32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1}


class LitePose(object):
    def __init__(self, cfg, width_mult=1.0, round_nearest=8, cfg_arch=None, storage=None):
        """``storage``: 'f32' (the reference's arithmetic) or 'bf16' (activations + folded weights in bf16,
        fp32 accumulation: the counterpart of the reference's reduced-precision switch ``cfg.FP16.ENABLED``,
        valid.py:152-153 -> fp16util.py:87-91 network_to_half, which is also the default when None)."""
        if width_mult != 1.0 or round_nearest != 8:
            raise ValueError('width_mult/round_nearest other than the defaults are not on the path')
        if storage is None:
            storage = 'bf16' if bool(cfg.FP16.ENABLED) else 'f32'
        if storage not in STORAGE:
            raise ValueError('storage must be one of %s' % sorted(STORAGE))
        self.storage = 'bf16' if STORAGE[storage] else 'f32'
        self._lib = nv.lib()
        self._arch = _arch_struct(cfg, cfg_arch)
        h = C.c_void_p()
        nv.check(self._lib.lp_net_create(C.byref(h), C.byref(self._arch)), 'lp_net_create')
        self._h = h
        self.final_channel = [int(self._arch.head_channels[i]) for i in range(self._arch.num_deconv - 1)]
        self._ws = None
        self._finalized = False
        self.training = False

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.lp_net_destroy(h)
            self._h = None

    # ---- nn.Module-shaped surface ------------------------------------------------
    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def keys(self):
        out = []
        shp = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self._lib.lp_net_num_keys(self._h)):
            k = self._lib.lp_net_key(self._h, i, shp, C.byref(nd))
            out.append((k.decode(), tuple(int(shp[d]) for d in range(nd.value))))
        return out

    def load_state_dict(self, state_dict, strict=True):
        expected = dict(self.keys())
        given = {}
        for k, v in state_dict.items():
            given[k[7:] if k.startswith('module.') else k] = v
        if strict:
            missing = [k for k in expected if k not in given]
            unexpected = [k for k in given if k not in expected]
            if missing or unexpected:
                raise RuntimeError('Error(s) in loading state_dict for LitePose: missing %s unexpected %s'
                                   % (missing[:5], unexpected[:5]))
        for k, v in given.items():
            if k not in expected:
                continue
            if k.endswith('num_batches_tracked'):
                nv.check(self._lib.lp_net_set_weight(self._h, k.encode(), None, None, 0), k)
                continue
            t = torch.as_tensor(v).detach().to('cpu', torch.float32).contiguous()
            if tuple(t.shape) != expected[k]:
                raise RuntimeError('size mismatch for %s: %s vs %s' % (k, tuple(t.shape), expected[k]))
            shp = (C.c_int64 * max(1, t.dim()))(*t.shape)
            nv.check(self._lib.lp_net_set_weight(self._h, k.encode(), C.c_void_p(t.data_ptr()), shp, t.dim()), k)
        nv.check(self._lib.lp_net_set_storage(self._h, STORAGE[self.storage]), 'lp_net_set_storage')
        nv.check(self._lib.lp_net_finalize(self._h, 1 if strict else 0), 'lp_net_finalize')
        self._finalized = True
        return self

    def state_dict(self):
        sd = OrderedDict()
        for k, shp in self.keys():
            if k.endswith('num_batches_tracked'):
                sd[k] = torch.zeros((), dtype=torch.int64)
                continue
            t = torch.empty(shp, dtype=torch.float32)
            nv.check(self._lib.lp_net_get_weight(self._h, k.encode(), C.c_void_p(t.data_ptr()), t.numel()), k)
            sd[k] = t
        return sd

    # ---- forward --------------------------------------------------------------------
    def _workspace(self, n, h, w, device):
        need = int(self._lib.lp_net_workspace_bytes(self._h, n, h, w))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws, need

    def forward_native(self, x, flip=0):
        """flip: 0 plain, 1 on flip(x,[3]), 2 both (outputs hold 2N images: plain then flipped)."""
        if not self._finalized:
            raise nv.LitePoseNativeError('load_state_dict() has not been called')
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected a float32 [N,3,H,W] tensor')
        x = x.contiguous()
        n, _, h, w = x.shape
        nb = 2 * n if flip == 2 else n
        out0 = torch.empty((nb, self.final_channel[0], h // 4, w // 4), dtype=torch.float32, device=x.device)
        out1 = torch.empty((nb, self.final_channel[1], h // 2, w // 2), dtype=torch.float32, device=x.device)
        ws, need = self._workspace(nb, h, w, x.device)
        nv.check(self._lib.lp_net_forward(self._h, nv.dptr(x), n, h, w, flip, nv.dptr(out0), nv.dptr(out1),
                                          nv.dptr(ws), need, nv.stream_ptr()), 'lp_net_forward')
        return [out0, out1]

    def forward(self, x):
        return self.forward_native(x, 0)

    __call__ = forward

    def tap(self, name):
        cnt = nv.check(self._lib.lp_net_tap(self._h, name.encode(), None, None), 'lp_net_tap')
        t = torch.empty(cnt, dtype=torch.float32, device=self._ws.device)
        nv.check(self._lib.lp_net_tap(self._h, name.encode(), nv.dptr(t), nv.stream_ptr()), 'lp_net_tap')
        return t

    def set_option(self, key, value):
        """Kernel-family switch of this net (include/litepose_amd.h: lp_net_set_option); returns the previous value."""
        prev = nv.check(self._lib.lp_net_get_option(self._h, key.encode()), 'lp_net_get_option')
        nv.check(self._lib.lp_net_set_option(self._h, key.encode(), int(value)), 'lp_net_set_option')
        return prev

    def get_option(self, key):
        return nv.check(self._lib.lp_net_get_option(self._h, key.encode()), 'lp_net_get_option')

    def set_profiling(self, enable):
        nv.check(self._lib.lp_net_set_profiling(self._h, 1 if enable else 0))

    def profile(self, cap=256, split=False, launches=False):
        """Per-launch (name|kernel, ms, algorithmic bytes, algorithmic FLOPs) of the last profiled forward;
        ``split=True`` appends the vector-pipe (depthwise) share of the FLOPs as a fifth field; ``launches=True`` appends
        the launch geometry (grid workgroups, threads per workgroup, LDS bytes, workgroups per CU by the occupancy query)
        as a sixth field (implies ``split``)."""
        names = ((C.c_char * 48) * cap)()
        ms = (C.c_float * cap)()
        by = (C.c_int64 * cap)()
        fl = (C.c_int64 * cap)()
        fv = (C.c_int64 * cap)()
        n = nv.check(self._lib.lp_net_profile2(self._h, names, ms, by, fl, fv, cap), 'lp_net_profile2')
        if launches:
            g, t, l, o = [(C.c_int32 * cap)() for _ in range(4)]
            nv.check(self._lib.lp_net_profile_launches(self._h, g, t, l, o, cap), 'lp_net_profile_launches')
            return [(names[i].value.decode(), float(ms[i]), int(by[i]), int(fl[i]), int(fv[i]),
                     (int(g[i]), int(t[i]), int(l[i]), int(o[i]))) for i in range(n)]
        if split:
            return [(names[i].value.decode(), float(ms[i]), int(by[i]), int(fl[i]), int(fv[i])) for i in range(n)]
        return [(names[i].value.decode(), float(ms[i]), int(by[i]), int(fl[i])) for i in range(n)]


def get_pose_net(cfg, is_train=False, cfg_arch=None, storage=None):
    """pose_mobilenet.py:158-176.  Pre-trained backbone loading (is_train and
    INIT_WEIGHTS) is a training feature and out of scope: weights always arrive through
    ``load_state_dict`` (valid.py:155-157).  ``storage`` (extension): see ``LitePose``."""
    return LitePose(cfg, cfg_arch=cfg_arch, storage=storage)
