"""lp_parse_mid (AE post-process straight from the stage-1-resolution merge, full-resolution maps never
written: the engine's default since round 5) and lp_parse_dm (heatmaps materialised by the det-only projection, tags
evaluated from ``mid``: the default of rounds 2-4, engine option ae='dm') against lp_tta_project + lp_parse on the same ``mid`` and against the oracle parser fed the
projected maps: records must be identical bit for bit.  Needs a real MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import group_ref, synth

pytestmark = pytest.mark.gpu


def _cfg(J):
    from litepose_amd import config
    cfg = config.get_cfg('coco' if J == 17 else 'crowd_pose')
    cfg.DATASET.NUM_JOINTS = cfg.MODEL.NUM_JOINTS = J
    return cfg


def _mid_scene(seed, N, J, h1, w1, T, people):
    """mid [N,4,J,h1,w1]: blob heatmaps (plain and a slightly different 'flip' copy), tag plateaus."""
    rng = np.random.default_rng(seed)
    mid = np.zeros((N, 4, J, h1, w1), np.float32)
    for n in range(N):
        d, t = synth.blob_scene(rng, J, h1, w1, 2, n_people=people[n % len(people)], sigma=2.0)
        mid[n, 0] = d
        mid[n, 1] = d * np.float32(0.97) + rng.uniform(0, 0.01, d.shape).astype(np.float32)
        mid[n, 2] = t[..., 0]
        mid[n, 3] = t[..., 1]
    return mid


def _run_both(mid_np, J, T, pcap=30, adjust=True, refine=True, nms_kernel=None, det_thr=None):
    from litepose_amd import _native as nv
    from litepose_amd.core import group
    lib = nv.lib()
    N, _, _, h1, w1 = mid_np.shape
    H, W = 2 * h1, 2 * w1
    cfg = _cfg(J)
    if nms_kernel is not None:
        cfg.TEST.NMS_KERNEL, cfg.TEST.NMS_PADDING = nms_kernel, nms_kernel // 2
    if det_thr is not None:
        cfg.TEST.DETECTION_THRESHOLD = det_thr
    p = group.HeatmapParser(cfg, person_capacity=pcap)
    mid = torch.from_numpy(mid_np).cuda()
    det = torch.empty((N, J, H, W), device='cuda')
    tag = torch.empty((N, J, H, W, T), device='cuda')
    nv.check(lib.lp_tta_project(nv.dptr(mid), N, J, h1, w1, H, W, T, nv.dptr(det), nv.dptr(tag), nv.stream_ptr()))
    need = int(lib.lp_parse_workspace_bytes(N, J, p.params.max_num_people, T, pcap))
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    # det-only projection: must write the very bits of the full projection's det
    det2 = torch.full((N, J, H, W), -7.0, device='cuda')
    nv.check(lib.lp_tta_project(nv.dptr(mid), N, J, h1, w1, H, W, T, nv.dptr(det2), None, nv.stream_ptr()))
    assert torch.equal(det, det2)
    out = []
    for which in ('maps', 'mid', 'dm'):
        ans = torch.zeros((N, pcap, J, 3 + T), device='cuda')
        cnt = torch.zeros((N,), dtype=torch.int32, device='cuda')
        sc = torch.zeros((N, pcap), device='cuda')
        if which == 'maps':
            nv.check(lib.lp_parse(nv.dptr(det), nv.dptr(tag), N, J, H, W, T, C.byref(p._q), pcap, int(adjust),
                                  int(refine), nv.dptr(ans), nv.dptr(cnt), nv.dptr(sc), nv.dptr(ws), need,
                                  nv.stream_ptr()), 'lp_parse')
        elif which == 'dm':
            nv.check(lib.lp_parse_dm(nv.dptr(det2), nv.dptr(mid), N, J, h1, w1, T, C.byref(p._q), pcap, int(adjust),
                                     int(refine), nv.dptr(ans), nv.dptr(cnt), nv.dptr(sc), nv.dptr(ws), need,
                                     nv.stream_ptr()), 'lp_parse_dm')
        else:
            nv.check(lib.lp_parse_mid(nv.dptr(mid), N, J, h1, w1, T, C.byref(p._q), pcap, int(adjust), int(refine),
                                      nv.dptr(ans), nv.dptr(cnt), nv.dptr(sc), nv.dptr(ws), need, nv.stream_ptr()),
                     'lp_parse_mid')
        torch.cuda.synchronize()
        out.append((ans.cpu().numpy(), cnt.cpu().numpy(), sc.cpu().numpy()))
    return out, det.cpu().numpy(), tag.cpu().numpy()


def _check(out, det, tag, J, pcap, adjust=True, refine=True, oracle=True):
    (a0, c0, s0), (a1, c1, s1), (a2, c2, s2) = out          # maps, mid, dm
    assert np.array_equal(c0, c1), (c0, c1)
    assert np.array_equal(c0, c2), (c0, c2)
    ora = group_ref.HeatmapParser(group_ref.Params(num_joints=J))
    persons = 0
    for n in range(len(c0)):
        k = min(int(c0[n]), pcap)
        assert np.array_equal(a0[n, :k], a1[n, :k]), (n, np.argwhere(a0[n, :k] != a1[n, :k])[:4])
        assert np.array_equal(s0[n, :k], s1[n, :k]), n
        assert np.array_equal(a0[n, :k], a2[n, :k]), ('dm', n, np.argwhere(a0[n, :k] != a2[n, :k])[:4])
        assert np.array_equal(s0[n, :k], s2[n, :k]), ('dm', n)
        if oracle:
            a, s = ora.parse_image(det[n], tag[n], adjust, refine)
            assert c1[n] == a.shape[0]
            assert np.array_equal(a1[n, :k], a[:k]) and np.array_equal(s1[n, :k], s[:k]), n
        persons += int(c0[n])
    return persons


@pytest.mark.parametrize('J,h1,w1,T,N', [(14, 128, 128, 2, 6), (17, 64, 64, 1, 3), (14, 48, 80, 2, 3),
                                         (14, 224, 224, 2, 1), (5, 20, 36, 2, 2)])
def test_parse_mid_equals_materialised_path_and_oracle(J, h1, w1, T, N):
    mid = _mid_scene(1000 + h1 + J, N, J, h1, w1, T, people=[3, 0, 9, 1, 14, 6])
    out, det, tag = _run_both(mid, J, T)
    assert _check(out, det, tag, J, 30) >= 3


@pytest.mark.parametrize('nms_kernel,det_thr', [(3, None), (7, None), (5, 0.0), (5, 0.5), (3, 0.30000001192092896)])
def test_parse_mid_nms_radii_and_thresholds(nms_kernel, det_thr):
    """Round 5's walk kernel has a radius-1 and a radius-2 variant (NMS_KERNEL 3 / 5; 7 falls back to the band kernel) and
    drops NMS survivors at or below DETECTION_THRESHOLD where they arise (match_by_tag never reads them).  All three AE
    paths must still agree record for record -- at threshold 0 (the prefilter degenerates to `> 0`), at a threshold that
    removes most blobs, and at one that is exactly a float (the `(double) v > thr` <-> `v > float_down(thr)` equivalence at
    its boundary).  The oracle comparison uses the same parameters."""
    J, T = 14, 2
    mid = _mid_scene(4242 + nms_kernel, 4, J, 96, 128, T, people=[7, 2, 11, 0])
    out, det, tag = _run_both(mid, J, T, nms_kernel=nms_kernel, det_thr=det_thr)
    (a0, c0, s0), (a1, c1, s1), (a2, c2, s2) = out
    assert np.array_equal(c0, c1) and np.array_equal(c0, c2), (c0, c1, c2)
    params = group_ref.Params(num_joints=J)
    params.nms_kernel, params.nms_padding = nms_kernel, nms_kernel // 2
    if det_thr is not None:
        params.detection_threshold = det_thr
    ora = group_ref.HeatmapParser(params)
    for n in range(len(c0)):
        k = min(int(c0[n]), 30)
        assert np.array_equal(a0[n, :k], a1[n, :k]) and np.array_equal(s0[n, :k], s1[n, :k]), ('mid', n)
        assert np.array_equal(a0[n, :k], a2[n, :k]) and np.array_equal(s0[n, :k], s2[n, :k]), ('dm', n)
        a, sc = ora.parse_image(det[n], tag[n], True, True)
        assert c1[n] == a.shape[0] and np.array_equal(a1[n, :k], a[:k]) and np.array_equal(s1[n, :k], sc[:k]), ('oracle', n)
    if det_thr != 0.5:
        assert int(c0.sum()) >= 5


def test_parse_mid_flags_and_small_capacity():
    mid = _mid_scene(77, 3, 14, 64, 64, 2, people=[5, 12, 2])
    for adj, ref in ((False, False), (True, False), (False, True)):
        out, det, tag = _run_both(mid, 14, 2, adjust=adj, refine=ref)
        _check(out, det, tag, 14, 30, adj, ref)
    out, det, tag = _run_both(mid, 14, 2, pcap=4)          # more persons than record slots
    _check(out, det, tag, 14, 4)
    assert out[1][1].max() > 4


def test_parse_mid_plateaus_take_the_exact_fallback():
    """Constant positive planes: every pixel survives the NMS, the key segments overflow and the kernel falls
    back to exact rounds over recomputed bands; all-zero and all-negative planes give no candidates."""
    J, h1, w1 = 14, 64, 64
    rng = np.random.default_rng(3)
    mid = np.zeros((4, 4, J, h1, w1), np.float32)
    mid[:, 2:] = rng.normal(size=(4, 2, J, h1, w1)).astype(np.float32)
    mid[0, 0, :3] = 0.5
    mid[0, 1, :3] = 0.5                                   # det == 0.5 everywhere on joints 0..2
    mid[1, :2] = -1.0
    mid[2, 0, 4, 10, 20] = 0.9
    mid[2, 1, 4, 10, 20] = 0.8
    # plateau ROWS whose value rises down the plane (joints 0..4): the bands overflow and every later survivor beats the
    # M-th best so far -- the insertion path of round 6's one-pass fallback, not just its early-out
    rows = np.where(np.arange(h1) % 3 == 0, 0.3 + 0.5 * np.arange(h1) / h1, 0.15).astype(np.float32)
    mid[3, 0, :5] = rows[:, None]
    mid[3, 1, :5] = rows[:, None]
    out, det, tag = _run_both(mid, J, 2)
    _check(out, det, tag, J, 30)


def test_parse_mid_saturated_batch_stays_within_a_small_multiple_of_a_normal_one():
    """with 4 bands per plane a saturated heatmap sent every band down the exact fallback, M selection rounds
    each re-evaluating the whole band (8.7 ms per 64 images against 0.49 for an ordinary scene; 13.5 ms on rising plateau
    rows).  Round 6's fallback is one pass with a sorted top-M across the lanes: 1.5 / 1.6 ms.  The bound here is loose (10 x an
    ordinary batch -- the grouping of 30 x 30 candidates per joint is most of what is left) but an M-round rescan fails it."""
    import time
    from litepose_amd import _native as nv
    from litepose_amd.core import group
    lib = nv.lib()
    N, J, h1, w1, T, pcap = 64, 14, 128, 128, 2, 30
    blob = _mid_scene(5, N, J, h1, w1, T, people=[8])
    sat = blob.copy()
    sat[:, :2] = 0.5
    stripes = blob.copy()
    rows = np.where(np.arange(h1) % 3 == 0, 0.3 + 0.5 * np.arange(h1) / h1, 0.15).astype(np.float32)
    stripes[:, 0] = stripes[:, 1] = rows[:, None]
    p = group.HeatmapParser(_cfg(J), person_capacity=pcap)
    need = int(lib.lp_parse_workspace_bytes(N, J, p.params.max_num_people, T, pcap))
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    ans = torch.zeros((N, pcap, J, 3 + T), device='cuda')
    cnt = torch.zeros((N,), dtype=torch.int32, device='cuda')
    sc = torch.zeros((N, pcap), device='cuda')

    def ms(mid_np):
        mid = torch.from_numpy(mid_np).cuda()
        f = lambda: nv.check(lib.lp_parse_mid(nv.dptr(mid), N, J, h1, w1, T, C.byref(p._q), pcap, 1, 1, nv.dptr(ans),
                                              nv.dptr(cnt), nv.dptr(sc), nv.dptr(ws), need, nv.stream_ptr()), 'lp_parse_mid')
        f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / 3 * 1e3

    base = ms(blob)
    for name, m in (('saturated', sat), ('stripes', stripes)):
        t = ms(m)
        assert t < 10 * base + 1.0, (name, t, base)


def test_engine_ae_paths_give_identical_records_and_maps():
    """PoseEngine on the default 'mid' path (round 5: nothing materialised, det and tags evaluated inside the walks), on
    the 'dm' path of rounds 2-4 (option ae='dm': det materialised, tags from mid) and on the reference-shaped materialised path
    (ae='maps'): same records, and the maps handed to the oracle by last_maps() are the same bits on all three."""
    import os
    from litepose_amd import arch_zoo, config, engine
    from oracle import inference_ref
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(config.get_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
    N, R = 6, 256
    x = synth.make_images(N, R, seed=9).cuda()
    off0, off1 = synth.lowres_offsets(19, N, 14, R)
    f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
    offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
    res = {}
    for mode, kw in (('mid', {}), ('dm', {'ae': 'dm'}), ('maps', {'ae': 'maps'})):
        eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30, **kw)       # the AE path is a constructor option
        assert eng._ae_path(R, R) == mode
        a, c, s = [t.clone() for t in eng.infer_batch(x, offsets=offs)]
        d, t = [m.clone() for m in eng.last_maps()]
        res[mode] = (a, c, s, d, t, eng._last[0][0])
    assert res['mid'][5] == 'mid' and res['maps'][5] == 'maps' and res['dm'][5] == 'mid'
    for mode in ('mid', 'dm'):
        for k in range(5):
            assert torch.equal(res[mode][k], res['maps'][k]), (mode, k)
    assert int(res['dm'][1].sum()) >= N
