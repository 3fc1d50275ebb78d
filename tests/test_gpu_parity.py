"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

from oracle import group_ref, inference_ref, net_ref, synth

pytestmark = pytest.mark.gpu


def _cfg(dataset='crowd_pose'):
    from litepose_amd import config
    return config.get_cfg(dataset)


def _model(arch_name, seed=1234, head_gain=1.0, cfg=None):
    from litepose_amd import arch_zoo
    from litepose_amd.models import pose_mobilenet
    arch = arch_zoo.get(arch_name)
    cfg = cfg or _cfg()
    sd = synth.make_state_dict(arch, seed=seed, head_gain=head_gain)
    m = pose_mobilenet.get_pose_net(cfg, is_train=False, cfg_arch=arch)
    m.load_state_dict(sd, strict=True)
    return m, arch, sd, cfg


# ------------------------------------------------------------------ network (P1)
# tolerance: north_star asks heatmaps within 1e-3 (fp32); asserted 2e-5 abs on outputs of range +-0.3
# (measured ~3e-7; a dropped bf16x3 cross term, 2^-16 relative, would show as ~5e-6 .. 1e-4)
NET_ATOL = 2e-5


@pytest.mark.parametrize('arch_name,N', [('search-XS', 2), ('search-S', 1)])
def test_net_vs_golden_64(golden, arch_name, N):
    m, arch, sd, cfg = _model(arch_name)
    x = synth.make_images(N, 64, seed=7)
    out = m(x.cuda())
    k = 'net_%s_64_' % arch_name
    for got, name in ((out[0], 'out0'), (out[1], 'out1')):
        np.testing.assert_allclose(got.cpu().numpy(), golden[k + name], rtol=0, atol=NET_ATOL)


def test_net_blockwise_vs_oracle_256():
    m, arch, sd, cfg = _model('search-XS')
    x = synth.make_images(2, 256, seed=3)
    taps = {}
    with torch.no_grad():
        ref = net_ref.forward(x, sd, arch, taps=taps)
    out = m(x.cuda())
    torch.cuda.synchronize()
    worst = 0.0
    for name in ['first'] + ['stage.%d.%d' % (s, b) for s, nb in enumerate((6, 8, 10, 10)) for b in range(nb)] \
            + ['deconv.0', 'deconv.1', 'deconv.2']:
        got = m.tap(name).cpu().numpy().reshape(taps[name].shape)
        err = float(np.abs(got - taps[name].numpy()).max()) / max(1.0, float(taps[name].abs().max()))
        worst = max(worst, err)
        assert err < 2e-5, (name, err)            # scaled by the tap's magnitude (trunk activations are O(10))
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=NET_ATOL)


def test_net_nonsquare_and_odd_planes():
    # 96x160 input: planes 48x80 .. 6x10 exercise ragged 16x16 tiles and pixel tails
    m, arch, sd, cfg = _model('search-XS')
    x = synth.make_images(3, 96, seed=5, w=160)
    with torch.no_grad():
        ref = net_ref.forward(x, sd, arch)
    out = m(x.cuda())
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=NET_ATOL)


@pytest.mark.parametrize('arch_name,R', [('search-M', 64), ('search-L', 64), ('prune-S', 96), ('search-S', 224)])
def test_other_archs_vs_oracle(arch_name, R):
    # wider nets: deconv filters 64/40 take the VALU deconv fallback, Cout > 32 the unfused blocks,
    # 224 -> 14x14 planes (H*W % 4 == 0 but 16-px tiles ragged), 96 -> 6x6 planes
    m, arch, sd, cfg = _model(arch_name)
    x = synth.make_images(2, R, seed=11)
    with torch.no_grad():
        ref = net_ref.forward(x, sd, arch)
    out = m(x.cuda())
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=NET_ATOL)


def test_batched_forward_is_bitwise_per_image():
    """P4 at the network level: every kernel choice depends on the layer shape only, and the
    image-paired depthwise never mixes its two images, so an odd batch (last image paired with
    itself), an even batch and single images give bit-identical outputs."""
    m, arch, sd, cfg = _model('search-XS')
    x = synth.make_images(5, 128, seed=77).cuda()
    full = [o.clone() for o in m(x)]
    for sl in (slice(0, 1), slice(1, 3), slice(2, 5), slice(4, 5)):
        part = m(x[sl].contiguous())
        for a, b in zip(part, full):
            assert torch.equal(a, b[sl]), sl
    both = m.forward_native(x[:3].contiguous(), flip=2)            # 3 plain + 3 mirrored in one pass
    flipped = m(torch.flip(x[:3], [3]).contiguous())
    for k in range(2):
        assert torch.equal(both[k][:3], full[k][:3])
        assert torch.equal(both[k][3:], flipped[k])


def test_flip_mode_matches_explicit_flip():
    m, arch, sd, cfg = _model('search-XS')
    x = synth.make_images(2, 128, seed=9).cuda()
    both = m.forward_native(x, flip=2)
    plain = m.forward_native(x, flip=0)
    flipped = m.forward_native(torch.flip(x, [3]).contiguous(), flip=0)
    only_f = m.forward_native(x, flip=1)
    for i in range(2):
        assert torch.equal(both[i][:2], plain[i])
        assert torch.equal(both[i][2:], flipped[i])
        assert torch.equal(only_f[i], flipped[i])


def test_state_dict_roundtrip_and_strict():
    m, arch, sd, cfg = _model('search-XS')
    back = m.state_dict()
    assert list(back.keys()) == list(sd.keys())
    for k in sd:
        if not k.endswith('num_batches_tracked'):
            assert torch.equal(back[k], sd[k]), k
    bad = dict(sd)
    bad.pop('first.0.0.weight')
    from litepose_amd.models import pose_mobilenet
    m2 = pose_mobilenet.get_pose_net(cfg, cfg_arch=arch)
    with pytest.raises(RuntimeError):
        m2.load_state_dict(bad, strict=True)
    with pytest.raises(Exception):
        m2(torch.zeros(1, 3, 64, 64).cuda())          # never finalized -> loud failure


# ------------------------------------------------------------------ TTA merge
def test_tta_merge_vs_oracle(golden):
    from litepose_amd.core import inference
    m, arch, sd, cfg = _model('search-XS')
    x = synth.make_images(2, 64, seed=7)
    outputs, heat, tags = inference.get_multi_stage_outputs(cfg, m, x.cuda(), True, True, (64, 64))
    fh, tl = inference.aggregate_results(cfg, 1, None, [], heat, tags)
    tg = torch.cat(tl, dim=4)
    np.testing.assert_allclose(fh.cpu().numpy(), golden['net_search-XS_64_heat'], rtol=0, atol=NET_ATOL)
    np.testing.assert_allclose(tg.cpu().numpy(), golden['net_search-XS_64_tags'], rtol=0, atol=NET_ATOL)
    # the merge alone, fed the device's own network outputs: tight tolerance
    n = 2
    outs = [outputs[0].cpu(), outputs[1].cpu()]
    outs_f = [outputs[2].cpu(), outputs[3].cpu()]
    ofh, otg = inference_ref.merge(outs, outs_f, inference_ref.TestCfg(), (64, 64))
    np.testing.assert_allclose(fh.cpu().numpy(), ofh.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(tg.cpu().numpy(), otg.numpy(), rtol=0, atol=2e-6)
    assert n == fh.shape[0]


@pytest.mark.parametrize('name', ['sq', 'rect', 'noflip', 'center', 'centerkeep', 'nop2i', 'nop2i_noflip', 'nop2i_first'])
def test_multiscale_aggregation_vs_reference(golden_ms, name):
    """valid.py:207-225 with TEST.SCALE_FACTOR of 2-3 entries: lp_tta_merge per scale (projected
    to the base size) + lp_maps_accumulate, against outputs of the real reference.  Round 6: also with
    TEST.PROJECT2IMAGE = False (inference.py:152 false arm, :180-189, :201-206): the maps of a scale stay at its stage-1
    resolution and aggregate_results resizes tags / flip-averaged heatmaps to the first scale's maps."""
    from litepose_amd.core import inference
    from test_oracle_pinning import _ms_case, _ms_center, _ms_p2i
    J, base, flip, per = _ms_case(golden_ms, name)
    cfg = _cfg('coco' if J in (17, 18) else 'crowd_pose')
    cfg.TEST.SCALE_FACTOR = [sc for sc, _, _ in per]
    cfg.TEST.FLIP_TEST = flip
    cfg.DATASET.WITH_CENTER, cfg.TEST.IGNORE_CENTER = _ms_center(golden_ms, name)
    cfg.DATASET.NUM_JOINTS = cfg.MODEL.NUM_JOINTS = J        # counts the centre joint (default.py:175)
    cfg.TEST.PROJECT2IMAGE = _ms_p2i(golden_ms, name)
    final, tags_list = None, []
    for sc, outs, outs_f in per:                       # stored in descending-scale order
        det, tag = inference.tta_merge(cfg, [o.cuda() for o in outs],
                                       [o.cuda() for o in outs_f] if flip else None,
                                       base if cfg.TEST.PROJECT2IMAGE else None)      # get_multi_stage_outputs' own rule
        final, tags_list = inference.aggregate_results(cfg, sc, final, tags_list,
                                                       inference._Merged([det]), inference._Merged([tag]))
    final = final / float(len(per))
    tags = torch.cat(tags_list, dim=4)
    np.testing.assert_allclose(final.cpu().numpy(), golden_ms[name + '_final'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tags.cpu().numpy(), golden_ms[name + '_tags'], rtol=0, atol=2e-6)


# ------------------------------------------------------------------ AE parser (P2: index-exact)
def _scenes(golden, seed):
    meta = golden['ae_%d_meta' % seed]
    J, R, T, n = [int(v) for v in meta[:4]]
    people = [int(v) for v in meta[4:]]
    sigma = 4.0 * R / 256.0 if R >= 128 else 2.0
    det, tag = synth.blob_batch(seed, n, J=J, H=R, W=R, T=T, people=people, sigma=sigma)
    return J, det, tag


def _parser(J):
    from litepose_amd.core import group
    return group.HeatmapParser(_cfg('coco' if J == 17 else 'crowd_pose'))


def _assert_same(got, ref_a, ref_s, what):
    a, s = got
    assert a.shape == ref_a.shape, (what, a.shape, ref_a.shape)
    assert np.array_equal(a, ref_a), (what, np.argwhere(a != ref_a)[:5])
    assert np.array_equal(s, ref_s), what


@pytest.mark.parametrize('seed', [101, 102, 103, 104])
def test_parse_matches_reference_goldens(golden, seed):
    J, det, tag = _scenes(golden, seed)
    res = _parser(J).parse_batch(det, tag)           # whole batch in one call (P4)
    for n in range(det.shape[0]):
        _assert_same(res[n], golden['ae_%d_%d_ans' % (seed, n)], golden['ae_%d_%d_scores' % (seed, n)],
                     (seed, n))


def test_parse_reference_shaped_api(golden):
    J, det, tag = _scenes(golden, 101)
    p = _parser(J)
    ans, scores = p.parse(torch.from_numpy(det[3:4]).cuda(), torch.from_numpy(tag[3:4]).cuda(), True, True)
    assert isinstance(ans, list) and len(ans) == 1
    assert np.array_equal(ans[0], golden['ae_101_3_ans'])
    assert np.array_equal(np.asarray(scores, np.float32), golden['ae_101_3_scores'])


def test_topk_matches_oracle_incl_tie_order(golden):
    # torch.topk's order among equal values is implementation-defined; oracle and device fix it to
    # (value desc, index asc), so this comparison covers plateaus / zero filler slots too (the reference
    # goldens can only be compared where values are unique and positive)
    J, det, tag = _scenes(golden, 104)
    p = _parser(J)
    tk = p.top_k(det, tag)
    ref = group_ref.top_k(det, tag, group_ref.Params(num_joints=J))
    for k in ('val_k', 'loc_k', 'tag_k'):
        assert np.array_equal(tk[k], ref[k]), k


def test_parse_stress_network_maps(golden):
    res = _parser(14).parse_batch(golden['stress_heat'][None], golden['stress_tags'][None])
    _assert_same(res[0], golden['stress_ans'], golden['stress_scores'], 'stress')


def test_parse_adjust_refine_flags(golden):
    J, det, tag = _scenes(golden, 104)
    ora = group_ref.HeatmapParser(group_ref.Params(num_joints=J))
    p = _parser(J)
    for adj, ref in ((False, False), (True, False), (False, True)):
        res = p.parse_batch(det[:3], tag[:3], adj, ref)
        for n in range(3):
            a, s = ora.parse_image(det[n], tag[n], adj, ref)
            _assert_same(res[n], a, s, (adj, ref, n))


def test_parse_edge_cases():
    p = _parser(14)
    ora = group_ref.HeatmapParser(group_ref.Params())
    rng = np.random.default_rng(5)
    H = W = 64
    # all-zero, all-negative, constant plateau (every pixel survives NMS), single pixel
    det = np.zeros((4, 14, H, W), np.float32)
    tag = rng.normal(size=(4, 14, H, W, 2)).astype(np.float32)
    det[1] = -1.0
    det[2, :3] = 0.5
    det[3, 0, 10, 20] = 0.9
    det[3, 5, 63, 63] = 0.7
    res = p.parse_batch(det, tag)
    for n in range(4):
        a, s = ora.parse_image(det[n], tag[n])
        _assert_same(res[n], a, s, ('edge', n))


@pytest.mark.parametrize('max_pts,levels', [(9, 6), (24, 3)])
def test_group_tie_breaking_matches_munkres(max_pts, levels):
    """Quantised values/tags force exact cost ties: exercises the munkres tie rules.  With few tag levels many unmatched
    candidates carry the KEY of an existing person (first tag component equal, second one far): the order-dependent slot
    re-use of group.py:80-92, which the kernel's parallel assignment must detect and hand to its sequential loop."""
    p = _parser(14)
    ora = group_ref.HeatmapParser(group_ref.Params())
    rng = np.random.default_rng(17)
    H = W = 48
    N = 12
    det = np.zeros((N, 14, H, W), np.float32)
    tag = np.zeros((N, 14, H, W, 2), np.float32)
    for n in range(N):
        npts = int(rng.integers(2, max_pts))
        for j in range(14):
            for k in range(npts):
                y, x = int(rng.integers(0, H // 6)) * 6 + 2, int(rng.integers(0, W // 6)) * 6 + 2
                det[n, j, y, x] = 0.25 * int(rng.integers(1, 4)) + 1e-3 * (y * W + x) / (H * W)
                tag[n, j, y, x] = (3.0 / levels) * rng.integers(0, levels, size=2)
    res = p.parse_batch(det, tag, True, False)
    for n in range(N):
        a, s = ora.parse_image(det[n], tag[n], True, False)
        _assert_same(res[n], a, s, ('ties', n))


# ------------------------------------------------------------------ end to end
def test_engine_e2e_device_maps_vs_oracle_parser():
    """P2 protocol: the reference-semantics parser is fed the bit-identical maps the device
    produced; keypoints must be identical.  Also checks batched == per-image (P4)."""
    from litepose_amd import arch_zoo, engine
    cfg = _cfg()
    arch = arch_zoo.get('search-XS')
    sd = synth.make_state_dict(arch, seed=1234)
    eng = engine.PoseEngine(cfg, arch, sd)
    N, R = 4, 256
    x = synth.make_images(N, R, seed=21).cuda()
    off0, off1 = synth.lowres_offsets(33, N, 14, R, people=[3, 0, 7, 12])
    f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
    offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
    ans, count, scores = eng.infer_batch(x, offsets=offs)
    det, tag = [t.cpu().numpy() for t in eng.last_maps()]
    ans, count, scores = ans.cpu().numpy(), count.cpu().numpy(), scores.cpu().numpy()
    ora = group_ref.HeatmapParser(group_ref.Params())
    total = 0
    for n in range(N):
        a, s = ora.parse_image(det[n], tag[n])
        assert count[n] == a.shape[0], (n, count[n], a.shape)
        assert np.array_equal(ans[n, :count[n]], a)        # identity back-projection on square input
        assert np.array_equal(scores[n, :count[n]], s)
        total += a.shape[0]
    assert total >= 10                                      # the scenes really contain people
    # conv path vs full CPU pipeline: heatmap error (P1) -- north_star bound 1e-3, held 2e-5
    with torch.no_grad():
        outs = net_ref.forward(x.cpu(), sd, arch)
        outs_f = net_ref.forward(torch.flip(x.cpu(), [3]), sd, arch)
        outs = [outs[0] + torch.from_numpy(off0), outs[1] + torch.from_numpy(off1)]
        outs_f = [outs_f[0] + torch.from_numpy(f0), outs_f[1] + torch.from_numpy(f1)]
        fh, tg = inference_ref.merge(outs, outs_f, inference_ref.TestCfg(), (R, R))
    assert float(np.abs(det - fh.numpy()).max()) < NET_ATOL
    assert float(np.abs(tag - tg.numpy()).max()) < NET_ATOL
    # batch-1 runs give the same records
    for n in (0, 2):
        o = (offs[0][[n, N + n]].contiguous(), offs[1][[n, N + n]].contiguous())
        a1, c1, s1 = eng.infer_batch(x[n:n + 1].contiguous(), offsets=o)
        assert int(c1[0]) == count[n]
        assert np.array_equal(a1[0, :count[n]].cpu().numpy(), ans[n, :count[n]])


def test_pipelined_submit_matches_infer_batch():
    """PoseEngine.submit (two lanes, batches in flight concurrently) returns the records of
    infer_batch, bitwise, for interleaved batches of different content."""
    from litepose_amd import arch_zoo, engine
    cfg = _cfg()
    arch = arch_zoo.get('search-XS')
    sd = synth.make_state_dict(arch, seed=1234)
    eng = engine.PoseEngine(cfg, arch, sd, person_capacity=64)
    R, N = 128, 4
    batches = []
    for k in range(3):
        x = synth.make_images(N, R, seed=40 + k).cuda()
        off0, off1 = synth.lowres_offsets(50 + k, N, 14, R, people=[2, 5, 0, 9])
        f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
        offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
        batches.append((x, offs))
    ref = []
    for x, offs in batches:
        a, c, s = eng.infer_batch(x, offsets=offs)
        ref.append((a.cpu().numpy().copy(), c.cpu().numpy().copy(), s.cpu().numpy().copy()))
    pend = [eng.submit(x, offsets=offs) for x, offs in batches[:2]]
    got = []
    for k in range(3):
        p = pend.pop(0)
        a, c, s = p.result()
        got.append((a.cpu().numpy().copy(), c.cpu().numpy().copy(), s.cpu().numpy().copy()))
        p.release()
        if k == 0:
            pend.append(eng.submit(*[batches[2][0]], offsets=batches[2][1]))
    for (a, c, s), (ra, rc, rs) in zip(got, ref):
        assert np.array_equal(c, rc)
        for n in range(N):
            assert np.array_equal(a[n, :c[n]], ra[n, :rc[n]])
            assert np.array_equal(s[n, :c[n]], rs[n, :rc[n]])
    assert sum(int(r[1].sum()) for r in ref) > 10


@pytest.mark.parametrize('arch_name', ['search-XS', 'search-S', 'search-M', 'search-L', 'prune-S', 'prune-M', 'prune-L'])
def test_every_published_arch_vs_reference_samples(golden_archs, arch_name):
    """The device network against outputs of the REAL reference module (tests/golden/gen_golden_archs.py) for all
    seven published architectures: 64x64, and 96x160 for the search-M / search-L families."""
    m, arch, sd, cfg = _model(arch_name)
    sizes = [(64, 64)] + ([(96, 160)] if arch_name in ('search-M', 'search-L') else [])
    for H, W in sizes:
        x = synth.make_images(1, H, seed=11, w=W)
        out = m(x.cuda())
        for k, t in enumerate(out):
            key = '%s_%dx%d_out%d' % (arch_name, H, W, k)
            assert tuple(t.shape) == tuple(golden_archs[key + '_shape'])
            np.testing.assert_allclose(t.cpu().numpy().reshape(-1)[::13], golden_archs[key + '_sample'],
                                       rtol=0, atol=NET_ATOL)
