"""The oracle vs the committed golden fixtures (outputs of the REAL reference code,
written by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_arch
from oracle import group_ref, inference_ref, munkres_ref, net_ref, spec, synth, transforms_ref


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)     # goldens were generated single-threaded (oneDNN order)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize('arch_name', ['search-XS', 'search-S'])
def test_net_and_merge_match_reference(golden, arch_name):
    arch = load_arch(arch_name)
    N = 2 if arch_name == 'search-XS' else 1
    sd = synth.make_state_dict(arch, seed=1234)
    x = synth.make_images(N, 64, seed=7)
    with torch.no_grad():
        out = net_ref.forward(x, sd, arch)
        fh, tags = inference_ref.run(lambda im: net_ref.forward(im, sd, arch), x, inference_ref.TestCfg())
    k = 'net_%s_64_' % arch_name
    # same ATen kernels, same thread count -> expected bit-identical; allow 1e-6 for other hosts
    for got, name in ((out[0], 'out0'), (out[1], 'out1'), (fh, 'heat'), (tags, 'tags')):
        np.testing.assert_allclose(got.numpy(), golden[k + name], rtol=0, atol=1e-6)


def test_net_128_stats(golden):
    arch = load_arch('search-XS')
    sd = synth.make_state_dict(arch, seed=1234)
    x = synth.make_images(1, 128, seed=7)
    with torch.no_grad():
        out = net_ref.forward(x, sd, arch)
    for t, nm in ((out[0], 'out0'), (out[1], 'out1')):
        np.testing.assert_allclose(t.numpy().reshape(-1)[::997], golden['net_search-XS_128_%s_sample' % nm],
                                   rtol=0, atol=1e-6)


def test_state_dict_scheme_counts():
    arch = load_arch('search-XS')
    shapes = spec.state_dict_shapes(arch)
    assert len(shapes) == 679
    n_params = sum(int(np.prod(s)) for k, s in shapes.items()
                   if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
    assert abs(n_params - 1.68e6) < 0.02e6


def test_munkres_matches_reference(golden):
    for i in range(60):
        m = golden['munkres_m%d' % i]
        p = golden['munkres_p%d' % i]
        assert munkres_ref.compute(m.copy()) == [tuple(x) for x in p.tolist()]


def _scenes(golden, seed):
    meta = golden['ae_%d_meta' % seed]
    J, R, T, n = [int(v) for v in meta[:4]]
    people = [int(v) for v in meta[4:]]
    sigma = 4.0 * R / 256.0 if R >= 128 else 2.0
    det, tag = synth.blob_batch(seed, n, J=J, H=R, W=R, T=T, people=people, sigma=sigma)
    return J, det, tag


@pytest.mark.parametrize('seed', [101, 103, 104])
def test_parser_matches_reference(golden, seed):
    J, det, tag = _scenes(golden, seed)
    parser = group_ref.HeatmapParser(group_ref.Params(num_joints=J))
    for n in range(det.shape[0]):
        a, s = parser.parse_image(det[n], tag[n])
        assert np.array_equal(a, golden['ae_%d_%d_ans' % (seed, n)])
        assert np.array_equal(s, golden['ae_%d_%d_scores' % (seed, n)])
    tk = group_ref.top_k(det[:1], tag[:1], parser.params)
    ref_v = golden['ae_%d_0_val_k' % seed]
    pos = ref_v > 0
    uniq = (ref_v[:, :, None] == ref_v[:, None, :]).sum(axis=2) == 1
    assert np.array_equal(ref_v * pos, tk['val_k'][0])
    assert np.array_equal(golden['ae_%d_0_loc_k' % seed][pos & uniq], tk['loc_k'][0][pos & uniq])
    assert np.array_equal(golden['ae_%d_0_tag_k' % seed][pos & uniq], tk['tag_k'][0][pos & uniq])


def test_parser_crowded_256(golden):
    J, det, tag = _scenes(golden, 102)
    parser = group_ref.HeatmapParser(group_ref.Params(num_joints=J))
    for n in range(det.shape[0]):
        a, s = parser.parse_image(det[n], tag[n])
        assert np.array_equal(a, golden['ae_102_%d_ans' % n])
        assert np.array_equal(s, golden['ae_102_%d_scores' % n])
    assert golden['ae_102_1_ans'].shape[0] > 30                  # persons beyond MAX_NUM_PEOPLE


def test_parser_stress_network_maps(golden):
    a, s = group_ref.HeatmapParser(group_ref.Params()).parse_image(golden['stress_heat'], golden['stress_tags'])
    assert np.array_equal(a, golden['stress_ans'])
    assert np.array_equal(s, golden['stress_scores'])


def test_mean_restatements_match_torch_and_numpy():
    rng = np.random.default_rng(0)
    for _ in range(300):
        n = int(rng.integers(1, 19))
        rows = (rng.normal(size=(n, 2)) * 3 + rng.normal() * 5).astype(np.float32)
        t = torch.cat([torch.from_numpy(rows[i:i + 1]) for i in range(n)], dim=0)
        assert np.array_equal(torch.mean(t, dim=0).numpy(), group_ref.torch_mean_dim0_f32(rows))
        assert np.array_equal(np.mean([r for r in rows], axis=0), group_ref._mean_rows_f32([r for r in rows]))
        J = int(rng.choice([5, 8, 9, 14, 17, 18]))
        a = rng.normal(size=(J, 5)).astype(np.float32)
        assert a[:, 2].mean() == group_ref.mean_strided_f32(a[:, 2])


def test_final_preds_identity_on_square_inputs():
    for R in (256, 448, 512):
        size, center, scale = transforms_ref.get_multi_scale_size((R, R), R, 1.0, 1.0)
        assert size == (R, R)
        person = np.random.default_rng(1).uniform(0, R, size=(14, 5)).astype(np.float32)
        out = transforms_ref.get_final_preds([[person]], center, scale, [R, R])
        np.testing.assert_allclose(out[0], person, rtol=0, atol=1e-4)


def _ms_case(g, name):
    J, bw, bh, flip, N = [int(v) for v in g[name + '_meta'][:5]]
    scales = [float(v) for v in g[name + '_scales']]
    per = []
    for idx, sc in enumerate(scales):
        outs = [torch.from_numpy(g['%s_s%d_f0_o%d' % (name, idx, k)]) for k in range(2)]
        outs_f = [torch.from_numpy(g['%s_s%d_f1_o%d' % (name, idx, k)]) for k in range(2)] if flip else None
        per.append((sc, outs, outs_f))
    return J, (bw, bh), bool(flip), per


def _ms_center(g, name):
    m = g[name + '_meta']
    return (bool(m[5]), bool(m[6])) if len(m) > 5 else (False, True)


def _ms_p2i(g, name):
    m = g[name + '_meta']
    return bool(m[7]) if len(m) > 7 else True


MS_CASES = ['sq', 'rect', 'noflip', 'center', 'centerkeep', 'nop2i', 'nop2i_noflip', 'nop2i_first']


@pytest.mark.parametrize('name', MS_CASES)
def test_multiscale_aggregation_matches_reference(golden_ms, name):
    """valid.py:207-225 multi-scale loop (the WITH_CENTER / IGNORE_CENTER channel handling of inference.py:148-150, and,
    round 6, TEST.PROJECT2IMAGE = False: inference.py:180-189, 201-206): oracle restatement == stored reference outputs,
    bitwise."""
    J, base, flip, per = _ms_case(golden_ms, name)
    wc, ic = _ms_center(golden_ms, name)
    tc = inference_ref.TestCfg(num_joints=J, dataset='coco_kpt' if J in (17, 18) else 'crowd_pose_kpt', flip_test=flip,
                               with_center=wc, ignore_center=ic, project2image=_ms_p2i(golden_ms, name))
    final, tags = inference_ref.merge_multiscale(list(reversed(per)), tc, base)   # any input order
    assert np.array_equal(final.numpy(), golden_ms[name + '_final'])
    assert np.array_equal(tags.numpy(), golden_ms[name + '_tags'])


# ---------------------------------------------------------------- pre-processing oracle (parity unpinned: no cv2)
def test_preprocess_oracle_self_consistency():
    """cv2 is absent, so the warp restatement can only be checked against itself: identity and
    integer shifts reproduce the image exactly, a smooth image stays within 2 grey levels of a float
    bilinear resampling, and ToTensor+Normalize equals the torch formula bit for bit."""
    from oracle import preprocess_ref as pr
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    assert np.array_equal(pr.warp_affine_u8(img, np.array([[1., 0, 0], [0, 1., 0]]), (53, 37)), img)
    o = pr.warp_affine_u8(img, np.array([[1., 0, 5], [0, 1., -3]]), (53, 37))
    assert np.array_equal(o[0:34, 5:53], img[3:37, 0:48]) and o[:, :5].max() == 0 and o[34:].max() == 0
    # smooth image, non-trivial scale: compare with float bilinear (taps outside -> 0)
    yy, xx = np.mgrid[0:120, 0:160].astype(np.float64)
    smooth = np.stack([127 + 100 * np.sin(xx / 23.0) * np.cos(yy / 17.0), xx * 255 / 159, yy * 255 / 119], 2)
    smooth = np.clip(np.rint(smooth), 0, 255).astype(np.uint8)
    res, center, scale = pr.resize_align_multi_scale(smooth, 128, 1.0, 1.0)
    assert res.shape == (128, 192, 3)
    m = pr.invert_affine(pr.get_affine_transform(center, scale, 0, (192, 128)))
    dy, dx = np.mgrid[0:128, 0:192].astype(np.float64)
    sx = m[0, 0] * dx + m[0, 1] * dy + m[0, 2]
    sy = m[1, 0] * dx + m[1, 1] * dy + m[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    pad = np.zeros((124, 164, 3))
    pad[2:122, 2:162] = smooth

    def tap(y, x):
        return pad[np.clip(y + 2, 0, 123), np.clip(x + 2, 0, 163)]
    ref = (1 - fy) * ((1 - fx) * tap(y0, x0) + fx * tap(y0, x0 + 1)) + fy * ((1 - fx) * tap(y0 + 1, x0) + fx * tap(y0 + 1, x0 + 1))
    inner = (sx > 1) & (sx < 158) & (sy > 1) & (sy < 118)
    assert np.abs(res.astype(np.float64) - ref)[inner].max() <= 2.0
    t = pr.to_tensor_normalize(img)
    tt = (torch.from_numpy(img.transpose(2, 0, 1).copy()).float().div(255)
          - torch.tensor([0.485, 0.456, 0.406])[:, None, None]) / torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    assert np.array_equal(t, tt.numpy())


ALL_ARCHS = ['search-XS', 'search-S', 'search-M', 'search-L', 'prune-S', 'prune-M', 'prune-L']


@pytest.mark.parametrize('arch_name', ALL_ARCHS)
def test_every_published_arch_matches_reference(golden_archs, arch_name):
    """Outputs of the real reference module for all seven mobile_configs/*.json (gen_golden_archs.py asserted
    bit-identity with net_ref while generating): stage outputs sampled every 13th value + whole-tensor sums."""
    arch = load_arch(arch_name)
    sd = synth.make_state_dict(arch, seed=1234)
    sizes = [(64, 64)] + ([(96, 160)] if arch_name in ('search-M', 'search-L') else [])
    for H, W in sizes:
        x = synth.make_images(1, H, seed=11, w=W)
        with torch.no_grad():
            out = net_ref.forward(x, sd, arch)
        for k, t in enumerate(out):
            key = '%s_%dx%d_out%d' % (arch_name, H, W, k)
            assert tuple(t.shape) == tuple(golden_archs[key + '_shape'])
            np.testing.assert_allclose(t.numpy().reshape(-1)[::13], golden_archs[key + '_sample'], rtol=0, atol=1e-6)
            a = t.numpy().astype(np.float64)
            st = np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.flat[::97].sum()])
            np.testing.assert_allclose(st, golden_archs[key + '_stats'], rtol=1e-6, atol=1e-5)
