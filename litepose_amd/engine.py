"""Batched fast path: images -> per-image keypoint records, entirely on the GPU.

``PoseEngine.infer_batch(images[N,3,H,W])`` = the body of the valid.py loop
(valid.py:195-245) generalised to a batch: network on the image and on its mirror
(one 2N launch sequence), flip-TTA merge + projection, NMS/top-k, tag grouping,
adjust/refine, back-projection.  No host synchronisation inside; results are
fixed-capacity records so that ranks can all-gather them (litepose_amd.parallel).
"""
import ctypes as C

import torch

from . import _native as nv
from .core import group as _group
from .core import inference as _inference
from .models import pose_mobilenet as _pm
from .utils import transforms as _tf


# Serving-schedule / AE-path options of a PoseEngine (round 6: constructor arguments -- the host layer reads no environment
# variable; ``options_from_env`` is the translation bench.py and tools/ apply for their LP_* experiment switches).
#
#   ae            'auto' | 'mid' | 'dm' | 'maps'   AE post-process (see _ae_path); 'auto' = 'mid' where its kernels apply
#   sched         'split' | 'lanes'                serving schedule of submit(): NET / AE stages on net_streams + ae_streams
#                                                  streams with ``lanes`` buffer sets | whole batches on free-running lanes
#   lanes         buffer sets (None: 4 for 'split', 2 for 'lanes')
#   net_streams, ae_streams, net_prio, ae_prio     streams of the split schedule and their HIP priorities (-1 = high)
#   split         'late' | 'early'                 'early' moves stage merge + projection from the NET to the AE stage
#   graph         True | False                     hipGraph replay of the stages (False: eager launches)
#   capture_mode  torch.cuda.graph capture_error_mode.  'thread_local' (default since round 3): only the capturing
#                 thread's own calls are checked.  Measured on ROCm 7.2 (tests/capture_probe.py): a second host thread
#                 polling events / streams during prepare() -- what torch.distributed's RCCL watchdog does -- leaves the
#                 captures intact (8.1 M foreign polls, none raised, all sets captured, records right), while under
#                 'global' the same polls raise hipErrorStreamCaptureUnsupported in the foreign thread, invalidate the
#                 capture and leave the stream unusable.
#   streams       internal fan-out of lp_net_forward inside submit() (None: 1 for 'split', 2 for 'lanes')
DEFAULT_OPTIONS = {'ae': 'auto', 'sched': 'split', 'lanes': None, 'net_streams': 2, 'ae_streams': 1, 'net_prio': -1,
                   'ae_prio': 0, 'split': 'late', 'graph': True, 'capture_mode': 'thread_local', 'streams': None}
_CHOICES = {'ae': ('auto', 'mid', 'dm', 'maps'), 'sched': ('split', 'lanes'), 'split': ('late', 'early'),
            'capture_mode': ('thread_local', 'global', 'relaxed')}


def make_options(options=None, **kw):
    """DEFAULT_OPTIONS overridden by ``options`` (a dict) and keyword arguments; unknown keys / values raise."""
    o = dict(DEFAULT_OPTIONS)
    for src in (options or {}), kw:
        for k, v in src.items():
            if k not in o:
                raise ValueError('unknown engine option %r (known: %s)' % (k, sorted(o)))
            if k in _CHOICES and v not in _CHOICES[k]:
                raise ValueError('engine option %s must be one of %s, not %r' % (k, _CHOICES[k], v))
            o[k] = v
    for k in ('net_streams', 'ae_streams'):
        if int(o[k]) < 1:
            raise ValueError('engine option %s must be >= 1' % k)
    return o


def options_from_env(env=None):
    """The LP_* experiment switches of bench.py / tools/ as an options dict (the engine itself never reads them):
    LP_AE=mid|dm|maps, LP_SCHED, LP_LANES, LP_NET_STREAMS, LP_AE_STREAMS, LP_NET_PRIO, LP_AE_PRIO, LP_SPLIT,
    LP_GRAPH=0, LP_CAPTURE_MODE, LP_STREAMS."""
    import os
    env = os.environ if env is None else env
    o = {}
    for key, name, conv in (('ae', 'LP_AE', str), ('sched', 'LP_SCHED', str), ('lanes', 'LP_LANES', int),
                            ('net_streams', 'LP_NET_STREAMS', int), ('ae_streams', 'LP_AE_STREAMS', int),
                            ('net_prio', 'LP_NET_PRIO', int), ('ae_prio', 'LP_AE_PRIO', int), ('split', 'LP_SPLIT', str),
                            ('capture_mode', 'LP_CAPTURE_MODE', str), ('streams', 'LP_STREAMS', int)):
        if env.get(name) not in (None, ''):
            o[key] = conv(env[name])
    if env.get('LP_GRAPH') not in (None, ''):
        o['graph'] = env['LP_GRAPH'] != '0'
    return o


_MAX_SHAPES = 3        # input shapes whose buffers (and captured graphs) stay resident per engine / buffer set


class PoseEngine(object):
    _buf_serial = 0

    def __init__(self, cfg, cfg_arch, state_dict, person_capacity=None, device=None, pipeline_halves=True,
                 ae_from_mid=False, storage=None, options=None, **option_kw):
        """``storage``: 'f32' | 'bf16' | None (= cfg.FP16.ENABLED, valid.py:152-153); see models.pose_mobilenet.
        ``options`` / keyword arguments: DEFAULT_OPTIONS above (AE path, serving schedule, graphs); ``ae_from_mid=True``
        is the older spelling of ``ae='mid'``."""
        self.cfg = cfg
        self.options = make_options(options, **option_kw)
        if ae_from_mid and self.options['ae'] == 'auto':
            self.options['ae'] = 'mid'
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.model = _pm.get_pose_net(cfg, is_train=False, cfg_arch=cfg_arch, storage=storage)
        self.model.load_state_dict(state_dict, strict=True)
        self.parser = _group.HeatmapParser(cfg, person_capacity=person_capacity)
        self.J = self.parser.params.num_joints              # joints in the merged maps / records
        self.Jn = int(cfg.DATASET.NUM_JOINTS)               # joints per network stage (incl. a centre joint)
        self.T = 2 if cfg.TEST.FLIP_TEST else 1
        self.pcap = self.parser.person_capacity
        self._bufs = {}
        self._lib = nv.lib()
        # two image halves on two HIP streams: the latency-bound AE kernels of one half (one wave per
        # image in the grouping, 1024-thread planes in NMS/refine) run under the other half's convs
        self.pipeline_halves = bool(pipeline_halves)
        self._side = None
        self._last = None
        self._lanes = None
        self._lane_next = 0
        self._stats = {'graph_replays': 0, 'graph_captures': 0, 'eager_stages': 0, 'capture_failures': 0}
        self._prepared = False
        self._in_prepare = False

    def _buffers(self, N, H, W):
        key = (N, H, W)
        b = self._bufs.get(key)
        if b is None:
            dev, J, T, pcap = self.device, self.J, self.T, self.pcap
            nb = 2 * N if self.cfg.TEST.FLIP_TEST else N
            b = {
                'out0': torch.empty((nb, 2 * self.Jn, H // 4, W // 4), dtype=torch.float32, device=dev),
                'out1': torch.empty((nb, self.Jn, H // 2, W // 2), dtype=torch.float32, device=dev),
                # full-resolution maps: only the det/tag path and last_maps() touch them (allocated on demand)
                'det': None, 'tag': None,
                'ans': torch.empty((N, pcap, J, 3 + T), dtype=torch.float32, device=dev),
                'count': torch.empty((N,), dtype=torch.int32, device=dev),
                'scores': torch.empty((N, pcap), dtype=torch.float32, device=dev),
            }
            m = self.model
            need = int(self._lib.lp_net_workspace_bytes(m._h, nb, H, W))
            b['net_ws'] = torch.empty(need, dtype=torch.uint8, device=dev)
            need_p = int(self._lib.lp_parse_workspace_bytes(N, J, self.parser.params.max_num_people, T, pcap))
            b['parse_ws'] = torch.empty(max(need_p, 256), dtype=torch.uint8, device=dev)
            # stage-1 intermediate of the TTA merge: one per engine instance (= per lane / per half), never
            # shared between streams
            need_t = int(self._lib.lp_tta_workspace_bytes(N, J, H // 2, W // 2))
            b['tta_ws'] = torch.empty(max(need_t, 256), dtype=torch.uint8, device=dev)
            # a few shapes stay resident (288 GB of HBM); every buffer dict carries a serial that is part of the
            # hipGraph keys of submit(): a graph holds raw pointers into ITS dict (and a reference to it, so the
            # memory outlives an eviction here) and can never be replayed against a newer dict of the same shape
            PoseEngine._buf_serial += 1
            b['serial'] = PoseEngine._buf_serial
            shapes = [k for k in self._bufs if isinstance(k, tuple) and len(k) == 3]
            while len(shapes) >= _MAX_SHAPES:
                self._bufs.pop(shapes.pop(0))
            self._bufs[key] = b
        elif list(self._bufs)[-1] != key:
            self._bufs[key] = self._bufs.pop(key)            # most recently used last
        return b

    def _forward_net(self, images, offsets=None, defer_offsets=False):
        """``defer_offsets``: do not add ``offsets`` here -- the caller hands them to the stage merge, which adds them
        as it reads the outputs (lp_tta_stage_add: bit-identical, one pass over the outputs less)."""
        cfg = self.cfg
        N, _, H, W = images.shape
        b = self._buffers(N, H, W)
        flip = 2 if cfg.TEST.FLIP_TEST else 0
        m = self.model
        nv.check(self._lib.lp_net_forward(m._h, nv.dptr(images), N, H, W, flip, nv.dptr(b['out0']),
                                          nv.dptr(b['out1']), nv.dptr(b['net_ws']), b['net_ws'].numel(),
                                          nv.stream_ptr()), 'lp_net_forward')
        m._ws = b['net_ws']
        if offsets is not None and not defer_offsets:
            b['out0'].add_(offsets[0])
            b['out1'].add_(offsets[1])
        outs = [b['out0'][:N], b['out1'][:N]]
        outs_f = [b['out0'][N:], b['out1'][N:]] if flip else None
        return b, outs, outs_f

    def _full_maps(self, b, N, H, W):
        if b['det'] is None:
            b['det'] = torch.empty((N, self.J, H, W), dtype=torch.float32, device=self.device)
        if b['tag'] is None:
            b['tag'] = torch.empty((N, self.J, H, W, self.T), dtype=torch.float32, device=self.device)
        return b['det'], b['tag']

    def forward_maps(self, images, offsets=None):
        """Network (+flip) + TTA merge with the full-resolution maps materialised like the reference does.
        ``offsets`` = optional (off0, off1) tensors of the network-output shapes added to the raw outputs
        (synthetic-scene injection used by the benchmark and the tests, SURVEY.md section 8d input 4).
        Returns (det, tag)."""
        cfg = self.cfg
        N, _, H, W = images.shape
        b, outs, outs_f = self._forward_net(images, offsets)
        if not cfg.TEST.PROJECT2IMAGE:
            raise NotImplementedError('PROJECT2IMAGE=False is not on the batched path')
        det, tag = self._full_maps(b, N, H, W)
        _inference.tta_merge(cfg, outs, outs_f, (W, H), det=det, tag=tag, ws=b['tta_ws'])
        return det, tag

    def forward_mid(self, images, offsets=None):
        """Network (+flip) + the stage merge only: returns the engine's ``mid`` buffer and its dims
        (N, J, h1, w1, T).  The fast path: ``parse_mid`` works on it directly."""
        N, _, H, W = images.shape
        defer = offsets is not None and self._can_defer(N, H, W, offsets)
        b, outs, outs_f = self._forward_net(images, offsets, defer)
        return (b['tta_ws'],) + _inference.tta_stage(self.cfg, outs, outs_f, b['tta_ws'],
                                                     add=offsets if defer else None)

    def _can_defer(self, N, H, W, offsets=None):
        """May ``offsets`` ride on the stage merge (lp_tta_stage_add reads them as fp32 arrays of exactly the stacked
        output shapes)?  Anything else -- broadcastable shapes, other dtypes, strided views -- takes the in-place
        ``add_`` after the network, with torch's own broadcasting / promotion rules."""
        if not _inference.stage_add_supported(N, self.J, H // 4, W // 4, H // 2, W // 2):
            return False
        if offsets is None:
            return True
        nf = 2 * N if self.cfg.TEST.FLIP_TEST else N
        Jn = int(self.cfg.DATASET.NUM_JOINTS)
        want = ((nf, 2 * Jn, H // 4, W // 4), (nf, Jn, H // 2, W // 2))
        return all(torch.is_tensor(o) and o.is_cuda and o.dtype == torch.float32 and o.is_contiguous()
                   and tuple(o.shape) == w for o, w in zip(offsets, want))

    def parse_mid(self, mid, N, J, h1, w1, T):
        cfg = self.cfg
        b = self._buffers(N, 2 * h1, 2 * w1)
        q = self.parser._q
        nv.check(self._lib.lp_parse_mid(nv.dptr(mid), N, J, h1, w1, T, C.byref(q), self.pcap,
                                        int(bool(cfg.TEST.ADJUST)), int(bool(cfg.TEST.REFINE)),
                                        nv.dptr(b['ans']), nv.dptr(b['count']), nv.dptr(b['scores']),
                                        nv.dptr(b['parse_ws']), b['parse_ws'].numel(), nv.stream_ptr()),
                 'lp_parse_mid')
        return b['ans'], b['count'], b['scores']

    def _ae_path(self, H, W):
        """Which AE post-process runs (identical records on all three, tests compare them):
          'mid'  (default since round 5 where its fast kernels apply: NMS_KERNEL 3 / 5, stage-1 width <= 512) NOTHING
                 materialised: lp_parse_mid evaluates det and tags from the stage-1-resolution merge inside the NMS
                 column walk and the refine walk (peaks_topk_walk_kernel, refine_dm_kernel<T, true>) -- no det-only
                 projection, no read-back: ~1 GB less HBM traffic per 64-image batch than 'dm'.  Needs
                 TEST.PROJECT2IMAGE with the exact x2 projection from the stage-1 resolution (every BASELINE config)
                 and TAG_PER_JOINT.
          'dm'   (option ae='dm'; the default of rounds 2-4, and still for NMS_KERNEL 7) heatmaps
                 materialised by the det-only projection, tags never: lp_tta_project(det only) + lp_parse_dm.
          'maps' (option ae='maps', and every other shape) the reference's full-resolution det + tag tensors.
        The path is an engine option (constructor), not an environment variable (round 6)."""
        p = self.parser.params
        # the gates of the native fast kernels (launch_tta_project(tag = NULL) exists only in the exact x2 kernel:
        # h1, w1 >= 2, N * J <= 65535 -- the last one is checked per call in _stage_merge)
        x2 = (bool(self.cfg.TEST.PROJECT2IMAGE) and W <= 1024 and W % 4 == 0 and H >= 4 and W >= 4
              and p.max_num_people <= 64
              and 3 <= int(self.cfg.TEST.NMS_KERNEL) <= 7 and bool(self.cfg.MODEL.TAG_PER_JOINT))
        if not x2:
            return 'maps'
        mode = self.options['ae']
        if mode == 'auto':
            # launch_peaks_topk_walk / the refine walk: NMS windows up to 5 x 5 (x2 already bounds the width)
            return 'mid' if int(self.cfg.TEST.NMS_KERNEL) <= 5 else 'dm'
        return mode

    def parse_dm(self, det, mid, N, J, h1, w1, T):
        cfg = self.cfg
        b = self._buffers(N, 2 * h1, 2 * w1)
        q = self.parser._q
        nv.check(self._lib.lp_parse_dm(nv.dptr(det), nv.dptr(mid), N, J, h1, w1, T, C.byref(q), self.pcap,
                                       int(bool(cfg.TEST.ADJUST)), int(bool(cfg.TEST.REFINE)),
                                       nv.dptr(b['ans']), nv.dptr(b['count']), nv.dptr(b['scores']),
                                       nv.dptr(b['parse_ws']), b['parse_ws'].numel(), nv.stream_ptr()),
                 'lp_parse_dm')
        return b['ans'], b['count'], b['scores']

    def parse_maps(self, det, tag):
        cfg = self.cfg
        N, J, H, W = det.shape
        b = self._buffers(N, H, W)
        q = self.parser._q
        nv.check(self._lib.lp_parse(nv.dptr(det), nv.dptr(tag), N, J, H, W, self.T, C.byref(q), self.pcap,
                                    int(bool(cfg.TEST.ADJUST)), int(bool(cfg.TEST.REFINE)),
                                    nv.dptr(b['ans']), nv.dptr(b['count']), nv.dptr(b['scores']),
                                    nv.dptr(b['parse_ws']), b['parse_ws'].numel(), nv.stream_ptr()),
                 'lp_parse')
        return b['ans'], b['count'], b['scores']

    def last_maps(self):
        """(det, tag) of the last infer_batch, whole batch (parity tests feed these to the oracle).  On the
        fast path the full-resolution maps were never written: they are projected from the engine's ``mid``
        buffer here, with the kernel lp_tta_merge itself uses."""
        parts = self._last
        if parts is None:
            raise RuntimeError('no batch has been processed yet')
        maps = []
        for p in parts:
            if p[0] == 'mid':
                _, eng, mid, N, J, h1, w1, T = p
                det, tag = eng._full_maps(eng._buffers(N, 2 * h1, 2 * w1), N, 2 * h1, 2 * w1)
                maps.append(_inference.tta_project(mid, N, J, h1, w1, (2 * w1, 2 * h1), T, det=det, tag=tag))
            else:
                maps.append(p[1:])
        if len(maps) == 1:
            return maps[0]
        return torch.cat([m[0] for m in maps]), torch.cat([m[1] for m in maps])

    def _stage_net(self, images, offsets, early=False):
        """First half of a batch: the network on the image and its mirror and (unless ``early``) the stage merge
        and the projection -- the chip-filling, bandwidth-heavy launches.  Returns the context of _stage_ae."""
        N, _, H, W = images.shape
        path = self._ae_path(H, W)
        if path == 'dm' and N * self.J > 65535:               # grid limit of the det-only projection
            path = 'maps'
        defer = offsets is not None and path != 'maps' and self._can_defer(N, H, W, offsets)
        b, outs, outs_f = self._forward_net(images, offsets, defer)
        add = offsets if defer else None
        if early:
            return (path, N, H, W, outs, outs_f, add)
        return self._stage_merge(path, N, H, W, outs, outs_f, add)

    def _stage_merge(self, path, N, H, W, outs, outs_f, add=None):
        cfg = self.cfg
        b = self._buffers(N, H, W)
        if path == 'maps':
            if not cfg.TEST.PROJECT2IMAGE:
                raise NotImplementedError('PROJECT2IMAGE=False is not on the batched path')
            det, tag = self._full_maps(b, N, H, W)
            _inference.tta_merge(cfg, outs, outs_f, (W, H), det=det, tag=tag, ws=b['tta_ws'])
            self._last = [('maps', det, tag)]
            return (path, N, H, W, det, tag)
        mid = b['tta_ws']
        _, J, h1, w1, T = _inference.tta_stage(cfg, outs, outs_f, mid, add=add)
        self._last = [('mid', self, mid, N, J, h1, w1, T)]         # last_maps(): maps re-projected on demand
        if path == 'dm':
            if b['det'] is None:
                b['det'] = torch.empty((N, J, H, W), dtype=torch.float32, device=self.device)
            _inference.tta_project(mid, N, J, h1, w1, (W, H), T, det=b['det'], det_only=True)
        return (path, N, H, W, mid, J, h1, w1, T)

    def _stage_ae(self, ctx, center, scale):
        """Second half: the AE post-process (NMS/top-k, grouping, adjust, refine, back-projection) -- latency-bound
        launches that fill a fraction of the chip."""
        path, N, H, W = ctx[:4]
        if isinstance(ctx[4], (list, tuple)):                 # early split: the merge runs here
            ctx = self._stage_merge(path, N, H, W, ctx[4], ctx[5], ctx[6])
        if path == 'maps':
            ans, count, scores = self.parse_maps(ctx[4], ctx[5])
        elif path == 'mid':
            ans, count, scores = self.parse_mid(ctx[4], N, *ctx[5:])
        else:
            b = self._buffers(N, H, W)
            ans, count, scores = self.parse_dm(b['det'], ctx[4], N, *ctx[5:])
        if center is None:
            # square network input of side INPUT_SIZE: get_multi_scale_size gives the identity
            (_, _), center, scale = _tf.get_multi_scale_size((H, W), min(H, W), 1.0, 1.0)
        _tf.final_preds_device(ans, count, center, scale, (W, H))
        return ans, count, scores

    def _infer_one(self, images, offsets, center, scale):
        return self._stage_ae(self._stage_net(images, offsets), center, scale)

    def infer_batch(self, images, offsets=None, center=None, scale=None):
        """images [N,3,H,W] float32 (normalised) on the GPU ->
        (kpts [N,pcap,J,3+T], count [N] int32, scores [N,pcap]); no host sync."""
        N = images.shape[0]
        if not self.pipeline_halves or N < 2 or N % 2:
            nv.check(self._lib.lp_net_set_streams(self.model._h, 2))
            return self._infer_one(images, offsets, center, scale)
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(2)]
            self._half = [PoseEngine.__new__(PoseEngine) for _ in range(2)]
            for hlf in self._half:               # same weights / parser, own buffers
                hlf.__dict__.update(self.__dict__)
                hlf._bufs = {}
                hlf.pipeline_halves = False
        nv.check(self._lib.lp_net_set_streams(self.model._h, 1))
        nh = N // 2
        J, T, pcap = self.J, self.T, self.pcap
        key = ('full', N)
        full = self._bufs.get(key)
        if full is None:
            dev = self.device
            full = (torch.empty((N, pcap, J, 3 + T), dtype=torch.float32, device=dev),
                    torch.empty((N,), dtype=torch.int32, device=dev),
                    torch.empty((N, pcap), dtype=torch.float32, device=dev))
            self._bufs[key] = full
        main = torch.cuda.current_stream()
        half_offs = None
        if offsets is not None:
            half_offs = [tuple(torch.cat([o[h * nh:(h + 1) * nh], o[N + h * nh:N + (h + 1) * nh]]) for o in offsets)
                         for h in range(2)]
        fork = torch.cuda.Event()
        fork.record(main)
        for h in range(2):
            sl = slice(h * nh, (h + 1) * nh)
            offs = None
            if offsets is not None:              # [plain N | mirrored N] -> this half's [plain | mirrored]
                # gathered on the caller's stream before the fork event: stream-ordered, no host sync
                offs = half_offs[h]
                for t in offs:                   # allocated on the caller's stream, consumed on the side stream
                    t.record_stream(self._side[h])
            with torch.cuda.stream(self._side[h]):
                self._side[h].wait_event(fork)
                a, c, s = self._half[h]._infer_one(images[sl], offs, center, scale)
                full[0][sl].copy_(a, non_blocking=True)
                full[1][sl].copy_(c, non_blocking=True)
                full[2][sl].copy_(s, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._side[h])
            main.wait_event(done)
        self._last = [h._last[0] for h in self._half]
        return full


    def submit(self, images, offsets=None, center=None, scale=None):
        """Software-pipelined serving.  Returns a ``PendingBatch``; inputs must stay alive/unchanged until
        ``result()`` has been waited on.

        Schedule (option sched='split', default): a batch is two stages, NET (network on image + mirror, stage merge,
        projection: chip-filling launches) and AE (NMS/top-k, grouping, adjust, refine: latency-bound launches on
        a fraction of the chip).  Batch k runs NET on net stream k % 2 and AE on the one AE stream, with buffer
        set k % 4: each net stream runs its networks back to back, so TWO networks are always in flight and the AE
        stages run underneath them.  Every stage is ONE chain of launches (no fan-out inside the
        network): a captured fork becomes extra graph-internal streams, and with more streams than hardware
        queues (4) a stream's event wait blocks the unrelated stream behind it in the same queue.
        sched='lanes' is the previous schedule, whole batches on two free-running lanes: the lanes drift into
        phase -- both in NET, then both in AE -- and the AE stage is exposed (tools/step_times.py).

        hipGraph: the launches of a stage are captured per buffer set the second time the set sees the same KEY
        -- input pointers and shapes (a serving loop that re-fills fixed staging buffers), centre / scale, the
        serial of the set's buffers and everything the host decides at capture time (AE path, ADJUST / REFINE /
        FLIP_TEST, the ``split`` option) -- and replayed as ONE graph launch afterwards, so the host cost per batch no longer
        scales with the launch count (8 ranks share the host's cores).  A set keeps the graphs of its last
        _MAX_SHAPES keys, each with a reference to the buffers it was captured on.  Kernel-family options
        (``model.set_option('mb16', 0)`` ..., lp_net_set_option) are baked into a captured graph: call
        ``reset_graphs()`` after changing one.
        Option ``graph=False`` disables."""
        self._ensure_lanes()
        lane = self._lanes[self._lane_next]
        self._lane_next = (self._lane_next + 1) % len(self._lanes)
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        nst = self.options['streams']
        nv.check(self._lib.lp_net_set_streams(self.model._h, int(nst) if nst else (1 if self._split else 2)))
        N, _, H, W = images.shape
        cfg = self.cfg
        early = self._split and self.options['split'] == 'early'
        key = (images.data_ptr(), tuple(images.shape),
               None if offsets is None else tuple((o.data_ptr(), tuple(o.shape)) for o in offsets),
               None if center is None else tuple(float(v) for v in center),
               None if scale is None else tuple(float(v) for v in scale),
               lane['eng']._buffers(N, H, W)['serial'], lane['eng']._ae_path(H, W), early,
               bool(cfg.TEST.ADJUST), bool(cfg.TEST.REFINE), bool(cfg.TEST.FLIP_TEST))
        if self._split:
            tensors, done = self._submit_split(lane, key, fork, images, offsets, center, scale, early)
        else:
            with torch.cuda.stream(lane['stream']):
                lane['stream'].wait_event(fork)
                if lane['consumed'] is not None:
                    lane['stream'].wait_event(lane['consumed'])
                ent = lane['graphs'].get(key) if self._use_graphs else None
                if ent is not None:
                    _touch(lane['graphs'], key)          # a hot graph must not be the LRU victim
                    ent['g'][0].replay()
                    tensors = ent['out']
                    self._stats['graph_replays'] += 1
                elif self._use_graphs and key in lane['seen'] and self._may_capture():
                    tensors = self._capture_lane(lane, key, images, offsets, center, scale)
                else:
                    tensors = lane['eng']._infer_one(images, offsets, center, scale)
                    self._stats['eager_stages'] += 1
                    _remember(lane['seen'], key)
                done = torch.cuda.Event()
                done.record(lane['stream'])
        self._last = lane['eng']._last
        return PendingBatch(lane, tensors, done)

    def _ensure_lanes(self):
        """Buffer sets and streams of the serving schedule, created on first use from ``self.options``."""
        if self._lanes is not None:
            return
        o = self.options
        self._split = o['sched'] != 'lanes'
        nl = max(1, int(o['lanes'] if o['lanes'] else (4 if self._split else 2)))
        self._lanes = [_make_lane(self) for _ in range(nl)]
        if self._split:                 # two net streams + one AE stream, shared by the buffer sets round-robin
            # HIP stream priorities (0 = default, -1 = high; the device's range is (0, -1)).  Round 5: the NET streams run at
            # HIGH priority -- the chip-filling network launches are never queued behind the AE stage's long, low-occupancy
            # workgroups (refine: 8 waves per CU for 0.36 ms), which have a whole step to finish: 3.037 -> 3.006 and
            # 3.011 -> 2.986 ms/step on two boxes, two repetitions each (profiles/r05_sched_experiments_box*.txt; AE high:
            # no change, a third NET stream on top: +3 %)
            ns = [torch.cuda.Stream(device=self.device, priority=int(o['net_prio'])) for _ in range(int(o['net_streams']))]
            as_ = [torch.cuda.Stream(device=self.device, priority=int(o['ae_prio'])) for _ in range(int(o['ae_streams']))]
            for i, ln in enumerate(self._lanes):
                ln['stream'], ln['ae_stream'] = ns[i % len(ns)], as_[i % len(as_)]
        self._use_graphs = bool(o['graph'])

    def _may_capture(self):
        """Lazy captures (second sight of a key inside a running serving loop) only in single-process runs: with
        torch.distributed up, collectives are in flight and its watchdog thread's event polls would invalidate the
        capture.  Multi-process runs capture in prepare() (nothing outstanding) -- or stay eager."""
        return self._in_prepare or _dist_world() <= 1

    def buffer_sets(self):
        """Number of buffer sets of the serving schedule = staging buffers a re-filling serving loop needs (a
        set's inputs must stay unchanged until its batch has been collected)."""
        self._ensure_lanes()
        return len(self._lanes)

    def prepare(self, images, offsets=None, center=None, scale=None):
        """One-time setup of a serving loop that re-fills FIXED staging buffers: allocates every buffer set and
        captures its stage graphs, which ``submit`` would otherwise do lazily over its first 2 x (buffer sets) calls
        (an eager pass that allocates, then the capture).  ``images`` is ONE staging tensor shared by all sets or a
        list of ``buffer_sets()`` tensors, set i being fed from ``images[i]`` (the order ``submit`` rotates in,
        starting with the next set to be used); ``offsets`` likewise one tuple or a list of tuples.  Contents do
        not matter.  Synchronises; afterwards every ``submit`` with these buffers is two graph launches."""
        self._ensure_lanes()
        self._prepared = True
        torch.cuda.synchronize()
        if _dist_world() > 1:
            # torch.distributed's watchdog thread polls the events of outstanding collectives.  What capture_probe.py
            # measured on ROCm 7.2 (header comment of this file): under 'thread_local', the engine's mode, such foreign
            # event / stream POLLS leave an open capture intact; under 'global' they invalidate it.  Other foreign HIP
            # calls (allocations, launches of a collective itself) were not probed, so belt and braces: after the
            # synchronize nothing is outstanding; give the watchdog one of its polling periods to retire what it still
            # lists.  Call prepare() before the first collective if you can.
            import time
            time.sleep(0.3)
        nl = len(self._lanes)
        imgs = list(images) if isinstance(images, (list, tuple)) else [images] * nl
        offs = list(offsets) if (isinstance(offsets, list)) else [offsets] * nl
        if len(imgs) != nl or len(offs) != nl:
            raise ValueError('prepare() needs one staging buffer (or one per buffer set: %d)' % nl)
        self._in_prepare = True
        try:
            for it in range(2 * nl):
                with self.submit(imgs[it % nl], offsets=offs[it % nl], center=center, scale=scale):
                    pass
            torch.cuda.synchronize()
        finally:
            self._in_prepare = False

    def reset_graphs(self):
        """Drop every captured graph (the next submits run eagerly once, then re-capture).  Needed after changing
        anything a capture bakes in that is not part of the key: the native LP_* experiment hooks."""
        torch.cuda.synchronize()
        for ln in (self._lanes or []):
            ln['graphs'].clear()
            ln['seen'].clear()

    def graph_stats(self):
        """{'use_graphs', 'captured_sets', 'graph_replays', 'graph_captures', 'eager_stages', 'capture_failures',
        'capture_mode'}: whether the serving loop really runs as graph replays (a failed capture drops the engine
        to eager launches for good; bench.py prints this so that a fallback is visible in the line)."""
        self._ensure_lanes()
        d = dict(self._stats)
        d['use_graphs'] = bool(self._use_graphs)
        d['captured_sets'] = sum(1 for ln in self._lanes if ln['graphs'])
        d['buffer_sets'] = len(self._lanes)
        d['capture_mode'] = self.options['capture_mode']
        return d

    def pipeline_depth(self):
        """How many submitted batches a serving loop should keep pending before it collects the oldest one
        (buffer sets - 2: a set is never re-used while its records may still be read).  Collecting earlier is
        correct but makes the next submit wait for the collected batch's AE stage through the caller's stream."""
        self._ensure_lanes()
        return max(1, len(self._lanes) - 2) if self._split else 1

    def _submit_split(self, lane, key, fork, images, offsets, center, scale, early):
        eng, ns, aes = lane['eng'], lane['stream'], lane['ae_stream']
        ent = lane['graphs'].get(key) if self._use_graphs else None
        replay = ent is not None
        capture = self._use_graphs and not replay and key in lane['seen'] and self._may_capture()
        if replay:
            _touch(lane['graphs'], key)
        ctx = None
        with torch.cuda.stream(ns):
            ns.wait_event(fork)
            if lane['ae_done'] is not None:          # the set's previous AE stage still reads mid / det
                ns.wait_event(lane['ae_done'])
            if replay:
                ent['g'][0].replay()
            elif capture:
                ent = self._capture_stage(None, 0, ns, lambda: eng._stage_net(images, offsets, early))
                capture = ent is not None
                if capture:
                    # the graph bakes in pointers of THESE buffers: keep them alive with it
                    ent['bufs'] = eng._buffers(images.shape[0], images.shape[2], images.shape[3])
                    ctx = ent['ctx']
            if not replay and not capture:
                ctx = eng._stage_net(images, offsets, early)
            net_done = torch.cuda.Event()
            net_done.record(ns)
        with torch.cuda.stream(aes):
            aes.wait_event(net_done)
            if lane['consumed'] is not None:         # the caller still reads the set's records
                aes.wait_event(lane['consumed'])
            if replay:
                ent['g'][1].replay()
                tensors = ent['out']
                self._stats['graph_replays'] += 1
            else:
                if capture and self._capture_stage(ent, 1, aes, lambda: eng._stage_ae(ctx, center, scale)) is not None:
                    tensors = ent['out']
                    _remember(lane['graphs'], key, ent)
                    self._stats['graph_captures'] += 1
                else:                                # first sight of the key, or a capture failed: eager
                    tensors = eng._stage_ae(ctx, center, scale)
                    _remember(lane['seen'], key)
                    self._stats['eager_stages'] += 1
            done = torch.cuda.Event()
            done.record(aes)
        lane['ae_done'] = done
        return tensors, done

    def _capture_stage(self, ent, idx, stream, fn):
        """Capture one stage of a buffer set into a hipGraph (buffers exist already: the set ran eagerly once)
        and launch it.  Returns the graph entry ({'g': [net, ae], 'ctx', 'out'}), or None after a failure: eager
        launches for good."""
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode=self.options['capture_mode']):
                out = fn()
            if idx == 0:
                ent = {'g': [g, None], 'ctx': out, 'out': None, 'bufs': None}
            else:
                ent['g'][1], ent['out'] = g, out
            g.replay()
            return ent
        except Exception as e:                           # capture is an optimisation, never a requirement
            self._capture_failed(e, stream)
            return None

    def _capture_lane(self, lane, key, images, offsets, center, scale):
        """Capture one batch of this lane into a hipGraph (buffers exist already: the lane ran eagerly
        once) and launch it.  Any failure falls back to eager launches for good."""
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=lane['stream'], capture_error_mode=self.options['capture_mode']):
                tensors = lane['eng']._infer_one(images, offsets, center, scale)
            _remember(lane['graphs'], key, {'g': [g], 'out': tensors, 'ctx': None,
                                            'bufs': lane['eng']._buffers(images.shape[0], images.shape[2],
                                                                         images.shape[3])})
            self._stats['graph_captures'] += 1
            g.replay()
            return tensors
        except Exception as e:                           # capture is an optimisation, never a requirement
            self._capture_failed(e, lane['stream'])
            return lane['eng']._infer_one(images, offsets, center, scale)

    def _capture_failed(self, exc, stream):
        """A capture raised (typically: another host thread made a HIP call while it was open).  Leave capture mode
        if the stream is still in it, clear the sticky error, drop every graph and stay with eager launches."""
        import warnings
        warnings.warn('hipGraph capture failed (%s); staying with eager launches' % (exc,))
        self._use_graphs = False
        self._stats['capture_failures'] += 1
        try:
            self._lib.lp_stream_abort_capture(C.c_void_p(stream.cuda_stream))
        except Exception:
            pass
        for ln in self._lanes:
            ln['graphs'].clear()
        try:
            torch.cuda.synchronize()
        except Exception:
            self._lib.lp_stream_abort_capture(C.c_void_p(stream.cuda_stream))


def _dist_world():
    try:
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    except Exception:
        return 1


def _remember(d, key, value=True):
    """Insert into a small most-recently-used dict (graphs / seen keys of a buffer set)."""
    d.pop(key, None)
    d[key] = value
    while len(d) > _MAX_SHAPES:
        d.pop(next(iter(d)))


def _touch(d, key):
    d[key] = d.pop(key)


class StagedLoader(object):
    """The two ends of the valid.py loop body that touch the host, for a serving loop over ``PoseEngine.submit``:
    uint8 HWC images in pinned host memory -> H2D -> ToTensor + Normalize on the device (valid.py:178-186,213;
    ``lp_preprocess_batch``) -> the buffer set's fp32 staging tensor, and the packed records of a collected batch ->
    pinned host memory (valid.py:232-245 reads them there).  One staging triple per buffer set, its own copy stream:
    the 12.5 MB of uint8 per 64 images cross PCIe under the previous batches' convolutions (the fp32 tensor would
    be 50 MB).  bench.py's ``value_with_io`` leg runs exactly this."""

    def __init__(self, engine, N, H, W, mean=None, std=None, own_stream=False):
        """``own_stream``: run the transfers on a stream of their own.  Off by default: the serving schedule already
        uses four streams (caller, two NET, one AE) and the device has four hardware queues -- a fifth stream shares
        a queue with one of them, and its copy waits behind that stream's whole network (round 3: 5.04 ms/step with
        a loader stream against the caller's stream)."""
        dev = engine.device
        self.nset = engine.buffer_sets()
        self.mean = tuple(mean) if mean is not None else _tf.IMAGENET_MEAN
        self.std = tuple(std) if std is not None else _tf.IMAGENET_STD
        self.host_u8 = [torch.empty((N, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(self.nset)]
        self.dev_u8 = [torch.empty((N, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(self.nset)]
        self.x = [torch.empty((N, 3, H, W), dtype=torch.float32, device=dev) for _ in range(self.nset)]
        self.stream = torch.cuda.Stream(device=dev) if own_stream else None
        self.host_rec = [None] * self.nset
        self.rec_done = [None] * self.nset
        self.ready = [None] * self.nset

    def start(self, i):
        """Begin H2D + normalisation of set i's images on the loader stream (asynchronous).  Call when set i's
        previous batch has been collected (its network no longer reads ``x[i]``) -- ideally one submit ahead, so
        that the transfer runs under the batch that is being submitted."""
        cur = torch.cuda.current_stream()
        if self.stream is None:                              # on the caller's stream: ordered by construction
            self.dev_u8[i].copy_(self.host_u8[i], non_blocking=True)
            _tf.normalize_batch_device(self.dev_u8[i], out=self.x[i], mean=self.mean, std=self.std)
            self.ready[i] = False
            return
        free = torch.cuda.Event()
        free.record(cur)                                     # everything queued so far may still read x[i]
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(free)
            self.dev_u8[i].copy_(self.host_u8[i], non_blocking=True)
            _tf.normalize_batch_device(self.dev_u8[i], out=self.x[i], mean=self.mean, std=self.std)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        self.ready[i] = ready

    def get(self, i):
        """``x[i]`` once the transfer started by ``start(i)`` is done (the current stream waits, not the host)."""
        if self.ready[i] is None:
            self.start(i)
        if self.ready[i] is not False:
            torch.cuda.current_stream().wait_event(self.ready[i])
        self.ready[i] = None
        return self.x[i]

    def load(self, i):
        """``start(i)`` + ``get(i)``: no prefetch."""
        self.start(i)
        return self.get(i)

    def store(self, i, kpts, count, scores):
        """Packed records of a collected batch -> pinned host buffer i (asynchronous; ``wait(i)`` before reading)."""
        from . import parallel as _par
        flat = _par.pack_records(kpts, count, scores)
        if self.host_rec[i] is None or self.host_rec[i].shape != flat.shape:
            self.host_rec[i] = torch.empty(flat.shape, dtype=flat.dtype).pin_memory()
        elif self.rec_done[i] is not None:
            self.rec_done[i].synchronize()                   # the previous copy into this host buffer
        self.host_rec[i].copy_(flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.rec_done[i] = ev
        return self.host_rec[i]

    def wait(self, i):
        if self.rec_done[i] is not None:
            self.rec_done[i].synchronize()
        return self.host_rec[i]


class PendingBatch(object):
    """Handle returned by ``PoseEngine.submit``: the batch is in flight on one of the engine's lanes."""

    def __init__(self, lane, tensors, done):
        self._lane, self._tensors, self._done = lane, tensors, done

    def result(self):
        """Make the current stream wait for the batch and return (kpts, count, scores).  The tensors
        are the buffer set's own: they stay valid until the set comes round again (4 sets: the fourth next
        ``submit``; sched='lanes': the second next)."""
        cur = torch.cuda.current_stream()
        cur.wait_event(self._done)
        # Safe default: the lane may re-use these buffers once everything queued on `cur` up to here is
        # done.  Consumers enqueued AFTER result() are covered by release(), which re-records the event.
        ev = torch.cuda.Event()
        ev.record(cur)
        self._lane['consumed'] = ev
        return self._tensors

    def release(self):
        """Call after the last consumer of the tensors has been enqueued on the current stream: the lane
        will not overwrite them before that consumer has run.  (Without it only work queued before
        result() is protected.)"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._lane['consumed'] = ev

    def __enter__(self):
        return self.result()

    def __exit__(self, *exc):
        self.release()
        return False


def _make_lane(engine):
    lane_eng = PoseEngine.__new__(PoseEngine)
    lane_eng.__dict__.update(engine.__dict__)
    lane_eng._bufs = {}
    lane_eng._side = None
    lane_eng._lanes = None
    lane_eng.pipeline_halves = False
    # graphs: key -> {'g': [NET graph, AE graph] (sched='lanes': [whole batch]), 'ctx', 'out', 'bufs'}, most recently
    # used last; seen: keys that ran eagerly once (their buffers exist: the next sight captures)
    return {'eng': lane_eng, 'stream': torch.cuda.Stream(device=engine.device), 'consumed': None,
            'graphs': {}, 'seen': {}, 'ae_stream': None, 'ae_done': None}


