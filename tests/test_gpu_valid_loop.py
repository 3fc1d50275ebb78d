"""The reference's evaluation loop body (valid.py:195-245) run on the drop-in modules, plus
randomised edge-shape checks of the AE entry points.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from oracle import group_ref, inference_ref, net_ref, synth, transforms_ref

pytestmark = pytest.mark.gpu


def test_valid_loop_body_drop_in():
    """Same call sequence as valid.py's loop (imports swapped), batch 1,
    against the oracle pipeline fed the device maps."""
    from litepose_amd import arch_zoo, config
    import litepose_amd.models as models
    from litepose_amd.core.inference import get_multi_stage_outputs, aggregate_results
    from litepose_amd.core.group import HeatmapParser
    from litepose_amd.utils.transforms import get_final_preds, get_multi_scale_size

    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(config.get_cfg('crowd_pose'), arch)
    model = models.pose_mobilenet.get_pose_net(cfg, is_train=True, cfg_arch=arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=6.0)       # noise peaks above the threshold
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    model.eval()
    parser = HeatmapParser(cfg)
    all_preds, all_scores = [], []
    R = cfg.DATASET.INPUT_SIZE
    for i in range(2):
        image = np.zeros((R, R, 3), np.uint8)                          # stands in for the decoded image
        base_size, center, scale = get_multi_scale_size(image, cfg.DATASET.INPUT_SIZE, 1.0, min(cfg.TEST.SCALE_FACTOR))
        final_heatmaps, tags_list = None, []
        for s in sorted(cfg.TEST.SCALE_FACTOR, reverse=True):
            image_resized = synth.make_images(1, R, seed=60 + i).cuda()   # ToTensor+Normalize output
            outputs, heatmaps, tags = get_multi_stage_outputs(cfg, model, image_resized, cfg.TEST.FLIP_TEST,
                                                              cfg.TEST.PROJECT2IMAGE, base_size)
            final_heatmaps, tags_list = aggregate_results(cfg, s, final_heatmaps, tags_list, heatmaps, tags)
        final_heatmaps = final_heatmaps / float(len(cfg.TEST.SCALE_FACTOR))
        tags = torch.cat(tags_list, dim=4)
        grouped, scores = parser.parse(final_heatmaps, tags, cfg.TEST.ADJUST, cfg.TEST.REFINE)
        final_results = get_final_preds(grouped, center, scale, [final_heatmaps.size(3), final_heatmaps.size(2)])
        all_preds.append(final_results)
        all_scores.append(scores)
        # oracle on the device's maps
        ora = group_ref.HeatmapParser(group_ref.Params())
        a, sc = ora.parse_image(final_heatmaps[0].cpu().numpy(), tags[0].cpu().numpy())
        ref = transforms_ref.get_final_preds([a], center, scale, [R, R])
        assert len(final_results) == len(ref) and len(ref) > 0
        for p, q in zip(final_results, ref):
            np.testing.assert_allclose(p, q, rtol=0, atol=1e-4)
        assert np.array_equal(np.asarray(scores, np.float32), sc)


def test_valid_loop_multiscale():
    """valid.py:195-245 with TEST.SCALE_FACTOR = [2, 1]: the network runs at two input sizes, every
    scale is projected to the scale-1 size, heatmaps are summed and tags come from scale 1."""
    from litepose_amd import arch_zoo, config
    import litepose_amd.models as models
    from litepose_amd.core.inference import get_multi_stage_outputs, aggregate_results
    from litepose_amd.core.group import HeatmapParser
    from litepose_amd.utils.transforms import get_multi_scale_size

    arch = dict(arch_zoo.get('search-XS'))
    arch['img_size'] = 64
    cfg = config.apply_arch(config.get_cfg('crowd_pose'), arch)
    cfg.TEST.SCALE_FACTOR = [1, 2]
    model = models.pose_mobilenet.get_pose_net(cfg, is_train=False, cfg_arch=arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=6.0)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    R = cfg.DATASET.INPUT_SIZE
    image = np.zeros((R, R, 3), np.uint8)
    base_size, center, scale = get_multi_scale_size(image, R, 1.0, min(cfg.TEST.SCALE_FACTOR))
    assert base_size == (R, R)
    final_heatmaps, tags_list, per = None, [], []
    for idx, s in enumerate(sorted(cfg.TEST.SCALE_FACTOR, reverse=True)):
        size_resized, _, _ = get_multi_scale_size(image, R, s, min(cfg.TEST.SCALE_FACTOR))
        assert size_resized == (R * s, R * s)
        x = synth.make_images(1, size_resized[0], seed=70 + idx)       # stands in for the warped image
        outputs, heatmaps, tags = get_multi_stage_outputs(cfg, model, x.cuda(), cfg.TEST.FLIP_TEST,
                                                          cfg.TEST.PROJECT2IMAGE, base_size)
        final_heatmaps, tags_list = aggregate_results(cfg, s, final_heatmaps, tags_list, heatmaps, tags)
        with torch.no_grad():
            per.append((s, net_ref.forward(x, sd, arch), net_ref.forward(torch.flip(x, [3]), sd, arch)))
    final_heatmaps = final_heatmaps / float(len(cfg.TEST.SCALE_FACTOR))
    tags = torch.cat(tags_list, dim=4)
    ofinal, otags = inference_ref.merge_multiscale(per, inference_ref.TestCfg(), base_size)
    assert tuple(final_heatmaps.shape) == tuple(ofinal.shape) and tuple(tags.shape) == tuple(otags.shape)
    np.testing.assert_allclose(final_heatmaps.cpu().numpy(), ofinal.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(tags.cpu().numpy(), otags.numpy(), rtol=0, atol=2e-5)
    grouped, scores = HeatmapParser(cfg).parse(final_heatmaps, tags, cfg.TEST.ADJUST, cfg.TEST.REFINE)
    a, sc = group_ref.HeatmapParser(group_ref.Params()).parse_image(final_heatmaps[0].cpu().numpy(),
                                                                    tags[0].cpu().numpy())
    assert a.shape[0] > 0 and np.array_equal(np.asarray(grouped[0], np.float32).reshape(a.shape), a)
    assert np.array_equal(np.asarray(scores, np.float32), sc)


@pytest.mark.parametrize('H,W,J,T', [(50, 70, 14, 2), (33, 129, 17, 1), (64, 64, 5, 2), (130, 66, 14, 1)])
def test_parser_random_maps_odd_shapes(H, W, J, T):
    """Non-multiple-of-4 widths take the scalar NMS / refine paths; J=17,T=1 is the COCO no-flip shape."""
    from litepose_amd import config
    from litepose_amd.core import group
    cfg = config.get_cfg('coco' if J == 17 else 'crowd_pose')
    cfg.DATASET.NUM_JOINTS = J
    cfg.MODEL.NUM_JOINTS = J
    p = group.HeatmapParser(cfg)
    params = group_ref.Params(num_joints=J)
    ora = group_ref.HeatmapParser(params)
    rng = np.random.default_rng(H * 1000 + W)
    N = 3
    det = np.zeros((N, J, H, W), np.float32)
    tag = np.zeros((N, J, H, W, T), np.float32)
    for n in range(N):
        d, t = synth.blob_scene(rng, J, H, W, T, n_people=int(rng.integers(0, 7)), sigma=2.0)
        det[n], tag[n] = d, t
    res = p.parse_batch(det, tag)
    tk = p.top_k(det, tag)
    rk = group_ref.top_k(det, tag, params)
    for k in ('val_k', 'loc_k', 'tag_k'):
        assert np.array_equal(tk[k], rk[k]), k
    for n in range(N):
        a, s = ora.parse_image(det[n], tag[n])
        assert res[n][0].shape == a.shape, (n, res[n][0].shape, a.shape)
        assert np.array_equal(res[n][0], a)
        assert np.array_equal(res[n][1], s)


def test_tta_merge_no_flip_and_odd_sizes():
    """flip_test off (T=1) and a projection that is not an exact x2 (generic bilinear path)."""
    from litepose_amd import config
    from litepose_amd.core import inference
    cfg = config.get_cfg()
    rng = np.random.default_rng(9)
    N, J = 2, 14
    out0 = torch.from_numpy(rng.normal(size=(N, 2 * J, 12, 20)).astype(np.float32))
    out1 = torch.from_numpy(rng.normal(size=(N, J, 24, 40)).astype(np.float32))
    out0f = torch.from_numpy(rng.normal(size=(N, 2 * J, 12, 20)).astype(np.float32))
    out1f = torch.from_numpy(rng.normal(size=(N, J, 24, 40)).astype(np.float32))
    for flip, size in ((True, (80, 48)), (False, (80, 48)), (True, (100, 60)), (True, (40, 24))):
        tc = inference_ref.TestCfg(flip_test=flip)
        ref_h, ref_t = inference_ref.merge([out0, out1], [out0f, out1f] if flip else None, tc, size)
        det, tag = inference.tta_merge(cfg, [out0.cuda(), out1.cuda()],
                                       [out0f.cuda(), out1f.cuda()] if flip else None, size)
        np.testing.assert_allclose(det.cpu().numpy(), ref_h.numpy(), rtol=0, atol=3e-6)
        np.testing.assert_allclose(tag.cpu().numpy(), ref_t.numpy(), rtol=0, atol=3e-6)


@pytest.mark.parametrize('hw,input_size,s,min_s', [((427, 640), 256, 1.0, 1.0), ((640, 427), 256, 1.0, 0.5),
                                                  ((100, 100), 128, 2.0, 1.0), ((333, 517), 448, 0.5, 0.5)])
def test_preprocess_matches_oracle(hw, input_size, s, min_s):
    """resize_align_multi_scale + ToTensor/Normalize as one kernel vs the NumPy restatement of the
    cv2 fixed-point warp: identical bytes and identical floats."""
    from litepose_amd.utils import transforms as T
    from oracle import preprocess_ref as pr
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    ref_u8, rc, rs = pr.resize_align_multi_scale(img, input_size, s, min_s)
    got, center, scale = T.resize_align_multi_scale(img, input_size, s, min_s)
    assert np.array_equal(center, rc) and np.array_equal(scale, rs)
    assert tuple(got.shape) == ref_u8.shape
    assert np.array_equal(got.cpu().numpy(), ref_u8)
    ten = T.ToTensorNormalize()(got)
    assert np.array_equal(ten.cpu().numpy(), pr.to_tensor_normalize(ref_u8))
    # the normalisation-only path (a warped image that did not come from our resize)
    ten2 = T.ToTensorNormalize()(torch.from_numpy(ref_u8).cuda())
    assert np.array_equal(ten2.cpu().numpy(), pr.to_tensor_normalize(ref_u8))


@pytest.mark.parametrize('flip', [True, False])
def test_stage_merge_with_additive_maps_is_the_in_place_add(flip):
    """lp_tta_stage_add (additive maps read together with the network outputs: the synthetic scenes of bench.py and
    of these tests, SURVEY 8d input 4) must give BITWISE the `mid` of adding the maps in place first and merging
    then; shapes the exact x2 merge does not cover are refused (the engine then adds in place)."""
    from litepose_amd import _native as nv
    from litepose_amd import config
    from litepose_amd.core import inference
    cfg = config.get_cfg()
    cfg.TEST.FLIP_TEST = flip
    rng = np.random.default_rng(5)
    N, J = 3, 14
    nf = 2 * N if flip else N
    out0 = torch.from_numpy(rng.normal(size=(nf, 2 * J, 16, 32)).astype(np.float32)).cuda()
    out1 = torch.from_numpy(rng.normal(size=(nf, J, 32, 64)).astype(np.float32)).cuda()
    a0 = torch.from_numpy(rng.normal(size=out0.shape).astype(np.float32)).cuda()
    a1 = torch.from_numpy(rng.normal(size=out1.shape).astype(np.float32)).cuda()
    need = int(nv.lib().lp_tta_workspace_bytes(N, J, 32, 64))
    mid_a = torch.zeros(need, dtype=torch.uint8, device='cuda')
    mid_b = torch.zeros(need, dtype=torch.uint8, device='cuda')
    assert inference.stage_add_supported(N, J, 16, 32, 32, 64)
    outs, outs_f = [out0[:N], out1[:N]], ([out0[N:], out1[N:]] if flip else None)
    inference.tta_stage(cfg, outs, outs_f, mid_a, add=(a0, a1))
    s0, s1 = out0 + a0, out1 + a1
    inference.tta_stage(cfg, [s0[:N], s1[:N]], [s0[N:], s1[N:]] if flip else None, mid_b)
    assert torch.equal(mid_a, mid_b)
    assert float(mid_a.view(torch.float32).abs().max()) > 0
    # 24-wide stage-1 maps: not a multiple of the 32-column block of the x2 kernel
    assert not inference.stage_add_supported(N, J, 16, 12, 32, 24)
    o0 = torch.zeros((nf, 2 * J, 16, 12), device='cuda')
    o1 = torch.zeros((nf, J, 32, 24), device='cuda')
    with pytest.raises(nv.LitePoseNativeError):
        inference.tta_stage(cfg, [o0[:N], o1[:N]], [o0[N:], o1[N:]] if flip else None, mid_a,
                            add=(torch.zeros_like(o0), torch.zeros_like(o1)))


def test_staged_loader_feeds_the_same_records():
    """The I/O-inclusive serving leg (bench.py value_with_io; valid.py:178-186,213,232-245): uint8 images in pinned
    host memory -> H2D -> lp_preprocess_batch -> PoseEngine.submit -> packed records in pinned host memory.  (i) the
    batched normalisation equals the oracle's ToTensor + Normalize per image, bit for bit, also with a warp; (ii) the
    records that come out of the loader loop are bitwise those of infer_batch on the oracle-normalised images, for
    every buffer set and through graph replay, with a different image batch per set."""
    from litepose_amd import arch_zoo, config, engine, parallel
    from litepose_amd.utils import transforms as T
    from oracle import preprocess_ref as pr
    rng = np.random.default_rng(77)
    N, R = 4, 128
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(config.get_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=6.0)
    eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
    nset = eng.buffer_sets()
    imgs = [rng.integers(0, 256, size=(N, R, R, 3), dtype=np.uint8) for _ in range(nset)]
    # (i) normalisation, identity and with the warp of resize_align_multi_scale
    x_ref = [np.ascontiguousarray(np.stack([pr.to_tensor_normalize(im[n]) for n in range(N)])) for im in imgs]
    got = T.normalize_batch_device(torch.from_numpy(imgs[0]).cuda())
    assert np.array_equal(got.cpu().numpy(), x_ref[0])
    big = rng.integers(0, 256, size=(3, 200, 300, 3), dtype=np.uint8)
    size, center, scale = T.get_multi_scale_size(big[0], 128, 1.0, 1.0)
    trans = T.get_affine_transform(center, scale, 0, size)
    warped = T.normalize_batch_device(torch.from_numpy(big).cuda(), trans=trans, size=size)
    for n in range(3):
        u8, _, _ = pr.resize_align_multi_scale(big[n], 128, 1.0, 1.0)
        assert np.array_equal(warped[n].cpu().numpy(), pr.to_tensor_normalize(u8)), n
    # (ii) the loader loop
    ref = []
    for i in range(nset):
        a, c, s = eng.infer_batch(torch.from_numpy(x_ref[i]).cuda())
        ref.append(parallel.pack_records(a, c, s).cpu().clone())
    assert sum(int(parallel.unpack_records(r, 30, 14, 5)[1].sum()) for r in ref) > 0, 'no persons: vacuous'
    loader = engine.StagedLoader(eng, N, R, R)
    for i in range(nset):
        loader.host_u8[i].copy_(torch.from_numpy(imgs[i]))
    eng.prepare(loader.x)
    depth = eng.pipeline_depth()
    pend = []
    for it in range(4 * nset):
        i = it % nset
        pend.append((i, eng.submit(loader.load(i) if it % 2 else loader.get(i))))
        if len(pend) > depth:
            j, h = pend.pop(0)
            loader.store(j, *h.result())
            h.release()
            assert torch.equal(loader.wait(j), ref[j]), (it, j)
    for j, h in pend:
        loader.store(j, *h.result())
        h.release()
        assert torch.equal(loader.wait(j), ref[j]), j
    st = eng.graph_stats()
    assert st['graph_replays'] >= 2 * nset and st['capture_failures'] == 0, st


def test_engine_with_center_ignore_center():
    """DATASET.WITH_CENTER + TEST.IGNORE_CENTER (inference.py:148-150, group.py:110-111): the network
    has 15 joints per stage, the merged maps and the records 14."""
    from litepose_amd import arch_zoo, config, engine
    arch = dict(arch_zoo.get('search-XS'))
    arch['img_size'] = 128
    cfg = config.enable_center(config.apply_arch(config.get_cfg('crowd_pose'), arch))
    assert cfg.DATASET.NUM_JOINTS == 15 and cfg.MODEL.NUM_JOINTS == 15
    from oracle import spec
    head = spec.HeadCfg(num_joints=15)
    sd = synth.make_state_dict(arch, head=head, seed=4321, head_gain=6.0)
    eng = engine.PoseEngine(cfg, arch, sd)
    N, R = 2, 128
    x = synth.make_images(N, R, seed=91)
    ans, count, scores = eng.infer_batch(x.cuda())
    det, tag = [t.cpu().numpy() for t in eng.last_maps()]
    assert det.shape == (N, 14, R, R) and tag.shape == (N, 14, R, R, 2) and ans.shape[2] == 14
    tc = inference_ref.TestCfg(num_joints=15, with_center=True, ignore_center=True)
    with torch.no_grad():
        outs = net_ref.forward(x, sd, arch, head=head)
        outs_f = net_ref.forward(torch.flip(x, [3]), sd, arch, head=head)
        fh, tg = inference_ref.merge(outs, outs_f, tc, (R, R))
    np.testing.assert_allclose(det, fh.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(tag, tg.numpy(), rtol=0, atol=2e-5)
    ora = group_ref.HeatmapParser(group_ref.Params(num_joints=15, with_center=True, ignore_center=True))
    cnt = count.cpu().numpy()
    a_dev = ans.cpu().numpy()
    people = 0
    for n in range(N):
        a, s = ora.parse_image(det[n], tag[n])
        assert cnt[n] == a.shape[0]
        assert np.array_equal(a_dev[n, :cnt[n]], a)
        people += a.shape[0]
    assert people > 0
