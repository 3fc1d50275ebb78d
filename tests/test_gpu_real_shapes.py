"""Parity at the BASELINE.json shapes themselves: XS@256 batch 64 through the
bench's own serving loop, S@448 / M@512 / L@512 network outputs, the fused 16x16-plane InvBottleneck
against the unfused kernel chain (bitwise), cross-stream workspace isolation.  Needs a real MI355X.

Tolerance: the network is fp32 (1x1 convs of the 16x16 planes as exact bf16x3 splits, dropped terms
<= 3*2^-24); outputs span about +-0.3, the oracle itself moves by 6e-8 with its thread count
(SURVEY.md 8d), the smoke case measures 2.4e-7.  Asserted: 2e-5 absolute on outputs and merged maps
(north_star budget 1e-3), 2e-5 * max(1, |tap|max) on block taps."""
import os

import numpy as np
import pytest
import torch

from oracle import group_ref, inference_ref, net_ref, synth

pytestmark = pytest.mark.gpu

OUT_ATOL = 2e-5


def _cfg():
    from litepose_amd import config
    return config.get_cfg('crowd_pose')


def _model(arch_name, seed=1234, head_gain=1.0):
    from litepose_amd import arch_zoo
    from litepose_amd.models import pose_mobilenet
    arch = arch_zoo.get(arch_name)
    sd = synth.make_state_dict(arch, seed=seed, head_gain=head_gain)
    m = pose_mobilenet.get_pose_net(_cfg(), is_train=False, cfg_arch=arch)
    m.load_state_dict(sd, strict=True)
    return m, arch, sd


def _offsets(seed, N, R, people=None):
    off0, off1 = synth.lowres_offsets(seed, N, 14, R, people=people)
    f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
    dev = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
    return (off0, off1, f0, f1), dev


# ------------------------------------------------------------------ fused 16x16-plane block
@pytest.mark.parametrize('arch_name,N', [('search-XS', 5), ('search-S', 2), ('search-L', 2)])
def test_mb16_fused_block_vs_unfused_chain_and_oracle(arch_name, N):
    """mb16_kernel (whole InvBottlenecks of a 16x16 plane, stages 3-4 at 256x256 input; both 1x1 on bf16x3 MFMAs)
    against the unfused pw3 -> dw_pair16 -> pw3 chain (lp_net_set_option 'mb16' = 0) on every block tap, and against
    the oracle.  Two fused forms: a RUN of same-shape residual blocks per launch (round 4, default: a block's output
    becomes the next block's input fragments with one v_permlane32_swap per register pair, the residual is rebuilt
    from the exact bf16 pieces the lane holds) and one block per launch ('mb16_run' = 0).  Both share fragment
    layouts and summation order with the chain: every tap and both outputs BITWISE."""
    m, arch, sd = _model(arch_name)
    x = synth.make_images(N, 256, seed=31).cuda()
    names = ['stage.%d.%d' % (s, b) for s in (2, 3) for b in range(10)]
    res = {}
    m.set_option('mb16_min', 0)                          # the fused form whatever the batch (round 6's small-batch rule off)
    try:
        for mode, (mb16, run) in (('run', (1, 1)), ('block', (1, 0)), ('chain', (0, 0))):
            m.set_option('mb16', mb16)
            m.set_option('mb16_run', run)
            m.set_profiling(True)
            out = [o.clone() for o in m(x)]
            kernels = [n.split('|')[1] for n, _, _, _ in m.profile()]
            m.set_profiling(False)
            res[mode] = (out, {k: m.tap(k).clone() for k in names}, kernels)
    finally:
        m.set_option('mb16', 1)
        m.set_option('mb16_run', 1)
        m.set_option('mb16_min', 48)
    nrun, nblk = res['run'][2].count('mb16_kernel'), res['block'][2].count('mb16_kernel')
    assert nblk >= 9 and 1 <= nrun < nblk, ('the fused kernel did not run / did not merge the runs', nrun, nblk)
    assert 'mb16_kernel' not in res['chain'][2]
    if arch_name == 'search-XS':
        assert (nrun, nblk) == (3, 19), (nrun, nblk)     # stage.2.1-9 | stage.3.0 | stage.3.1-9
    for mode in ('run', 'block'):
        diff = [k for k in names if not torch.equal(res[mode][1][k], res['chain'][1][k])]
        assert not diff, ('mb16_kernel must be bitwise the unfused chain', mode, diff)
        for a, b in zip(res[mode][0], res['chain'][0]):
            assert torch.equal(a, b), mode
    with torch.no_grad():
        ref = net_ref.forward(x.cpu(), sd, arch)
    for a, b in zip(res['run'][0], ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=OUT_ATOL)


def test_small_batch_rule_keeps_mb16_for_full_batches_only():
    """Round 6 (the reference evaluates at batch 1, valid.py:195-196): mb16_kernel is one workgroup per
    image, so a launch of a few images leaves the chip empty for the 1.13 ms the kernel takes whatever the batch (batch 1:
    network 1.67 ms with it, 1.22 ms on the pw3 / dw_pair16 / pw3 chain; profiles/r06_small_batch_per_launch.txt).  Option
    "mb16_min" (default 48 images per launch, mirrored ones included) routes smaller launches to the chain -- the same bits
    (P4: batched == batch-1), asserted here for batch 1 and 8 with the mirrored pass, and a full batch still takes the
    fused kernel."""
    m, arch, sd = _model('search-XS')
    assert m.get_option('mb16_min') == 48
    for N, want_fused in ((1, False), (8, False), (24, True)):
        x = synth.make_images(N, 256, seed=61 + N).cuda()
        res = {}
        for mn in (48, 0):
            m.set_option('mb16_min', mn)
            try:
                m.set_profiling(True)
                out = [o.clone() for o in m.forward_native(x, 2)]
                res[mn] = (out, [n.split('|')[1] for n, _, _, _ in m.profile()])
                m.set_profiling(False)
            finally:
                m.set_option('mb16_min', 48)
        assert ('mb16_kernel' in res[48][1]) == want_fused, (N, res[48][1])
        assert 'mb16_kernel' in res[0][1]
        for a, b in zip(res[48][0], res[0][0]):
            assert torch.equal(a, b), N


@pytest.mark.parametrize('arch_name,R,N', [('search-XS', 256, 1), ('search-XS', 256, 3), ('search-S', 224, 2), ('search-M', 192, 1)])
def test_pw3d_deep_prefetch_form_is_bitwise_pw3(arch_name, R, N):
    """Round 6 (batch-1 latency, valid.py:195-196): small launches of the bf16x3 1x1 (<= 8192 pixels) run as pw3d_kernel --
    loads four k-steps ahead, 32 pixels per wave -- because at a few images nothing else hides pw3_kernel's one-step-ahead
    loads (16 us for a K = 480 project of two images).  Same fragments, same six products per k-step, k-steps in order: every
    network output must be bit-identical with option "pw3d" = 0 (never), 1 (the size rule) and 2 (always), which is what keeps
    P4 (batched == per-image, bitwise) independent of the form a launch takes."""
    m, arch, sd = _model(arch_name)
    x = synth.make_images(N, R, seed=71 + N).cuda()
    assert m.get_option('pw3d') == 1
    res = {}
    for mode in (0, 1, 2):
        m.set_option('pw3d', mode)
        try:
            m.set_profiling(True)
            out = [o.clone() for o in m.forward_native(x, 2)]
            res[mode] = (out, [n.split('|')[1] for n, _, _, _ in m.profile()])
            m.set_profiling(False)
        finally:
            m.set_option('pw3d', 1)
    assert 'pw3d_kernel' not in res[0][1] and 'pw3_kernel' in res[0][1], res[0][1]
    assert 'pw3d_kernel' in res[2][1] and 'pw3_kernel' not in res[2][1], res[2][1]
    assert 'pw3d_kernel' in res[1][1], res[1][1]             # these launches are small: the rule takes the new form
    for mode in (1, 2):
        for a, b in zip(res[0][0], res[mode][0]):
            assert torch.equal(a, b), (mode, arch_name)


@pytest.mark.parametrize('arch_name,H,W,N', [('search-XS', 256, 256, 3), ('search-XS', 96, 160, 2),
                                             ('search-S', 448, 448, 1), ('search-XS', 80, 48, 2)])
def test_mbt_tiled_fused_block_vs_previous_kernels_and_oracle(arch_name, H, W, N):
    """mbt_kernel (round 3: whole InvBottleneck per 16x16 output tile, 8 waves, px-split bf16x3 projection;
    mbtile_kernels.hip) against what runs without it -- option "mbt" = 0: mbconv2_kernel (16-filter blocks) and the
    unfused chain (32-filter blocks, stride-2 first blocks: mbt_s2_kernel) -- on every block
    tap of stages 1-2 and on the stage-3 entry block, ragged tiles and image borders included (96x160 and 80x48
    inputs: 24x40 / 12x20 / 20x12 / 10x6 planes), and the outputs against the oracle.  "mbt" = 2 routes the
    16-filter blocks through it as well.  All forms are fp32-exact products with fp32 accumulation in different
    orders: a few ulp of the tap's magnitude."""
    m, arch, sd = _model(arch_name)
    x = synth.make_images(N, H, seed=33, w=W).cuda()
    names = ['stage.%d.%d' % (s, b) for s, nb in ((0, 6), (1, 8)) for b in range(nb)] + ['stage.2.0']
    res = {}
    for mode in ('2', '1', '0'):
        m.set_option('mbt', int(mode))
        try:
            m.set_profiling(True)
            out = [o.clone() for o in m(x)]
            kernels = [n.split('|')[1] for n, _, _, _ in m.profile()]
            m.set_profiling(False)
            res[mode] = (out, {k: m.tap(k).clone() for k in names}, kernels)
        finally:
            m.set_option('mbt', 1)
    assert 'mbt_kernel' not in res['0'][2] and 'mbt_s2_kernel' not in res['0'][2]
    ns2 = res['1'][2].count('mbt_s2_kernel')
    if max(H, W) >= 256:
        assert ns2 >= 2, ('the stride-2 first blocks did not take mbt_s2_kernel', res['1'][2])
    n2, n1 = res['2'][2].count('mbt_kernel'), res['1'][2].count('mbt_kernel')
    print('%s %dx%d: mbt launches with mbt = 2: %d, default: %d' % (arch_name, H, W, n2, n1))
    if max(H, W) >= 256:
        assert n2 >= 12 and n1 >= 7 and n2 > n1, (n2, n1)          # stride-1 blocks of stages 1-2: 5 + 7
    worst = 0.0
    for mode in ('2', '1'):
        for k in names:
            a, b = res[mode][1][k], res['0'][1][k]
            rel = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
            worst = max(worst, rel)
            assert rel < 2e-6, (mode, k, rel)
        for a, b in zip(res[mode][0], res['0'][0]):
            assert float((a - b).abs().max()) < 2e-6, mode
    print('worst scaled tap difference %.2e' % worst)
    with torch.no_grad():
        ref = net_ref.forward(x.cpu(), sd, arch)
    for mode in ('2', '1'):
        for a, b in zip(res[mode][0], ref):
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=OUT_ATOL)


# ------------------------------------------------------------------ BASELINE config 2/3: XS@256 b64
def test_xs256_batch64_bench_settings_vs_oracle():
    """The bench's exact serving configuration: XS@256, 64 images + 64 mirrored, pcap 30,
    head_gain 0.25, PoseEngine.submit (four buffer sets on two NET + two AE streams, two internal streams)."""
    from litepose_amd import arch_zoo, config, engine
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
    eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
    N, R = 64, 256
    x = synth.make_images(N, R, seed=100)
    (off0, off1, f0, f1), offs = _offsets(200, N, R)
    xd = x.cuda()
    # six submits, at most pipeline_depth + 1 in flight: every buffer set is exercised and two are re-used
    pend = [eng.submit(xd, offsets=offs), eng.submit(xd, offsets=offs)]
    first = pend.pop(0)
    a0, c0, s0 = [t.clone() for t in first.result()]
    first.release()
    outs = []
    for _ in range(4):
        pend.append(eng.submit(xd, offsets=offs))
        if len(pend) > eng.pipeline_depth():
            with pend.pop(0) as (a, c, s):
                outs.append((a.clone(), c.clone(), s.clone()))
    for p in pend:
        with p as (a, c, s):
            outs.append((a.clone(), c.clone(), s.clone()))
    det, tag = [t.cpu().numpy() for t in eng.last_maps()]
    for a, c, s in outs:                      # same input -> same records on either lane
        assert torch.equal(a, a0) and torch.equal(c, c0) and torch.equal(s, s0)
    ans, count, scores = a0.cpu().numpy(), c0.cpu().numpy(), s0.cpu().numpy()
    # P1: merged maps vs the full CPU oracle pipeline
    with torch.no_grad():
        o = net_ref.forward(x, sd, arch)
        of = net_ref.forward(torch.flip(x, [3]), sd, arch)
        o = [o[0] + torch.from_numpy(off0), o[1] + torch.from_numpy(off1)]
        of = [of[0] + torch.from_numpy(f0), of[1] + torch.from_numpy(f1)]
        fh, tg = inference_ref.merge(o, of, inference_ref.TestCfg(), (R, R))
    err_h = float(np.abs(det - fh.numpy()).max())
    err_t = float(np.abs(tag - tg.numpy()).max())
    print('XS@256 b64: heatmap max-abs err %.2e, tag %.2e' % (err_h, err_t))
    assert err_h < OUT_ATOL and err_t < OUT_ATOL
    # P2: records bit-exact against the reference-semantics parser on the device maps
    ora = group_ref.HeatmapParser(group_ref.Params())
    persons = 0
    for n in range(N):
        a, s = ora.parse_image(det[n], tag[n])
        assert count[n] == a.shape[0], (n, count[n], a.shape)
        k = min(int(count[n]), 30)
        assert np.array_equal(ans[n, :k], a[:k]) and np.array_equal(scores[n, :k], s[:k]), n
        persons += a.shape[0]
    assert persons >= N                       # 1..10 people per image
    # P4: batch-1 runs of a sample of the 64 images are bitwise the batched result
    for n in (0, 17, 63):
        o1 = (offs[0][[n, N + n]].contiguous(), offs[1][[n, N + n]].contiguous())
        a1, c1, s1 = eng.infer_batch(xd[n:n + 1].contiguous(), offsets=o1)
        d1, t1 = eng.last_maps()
        assert np.array_equal(d1[0].cpu().numpy(), det[n]) and np.array_equal(t1[0].cpu().numpy(), tag[n]), n
        assert int(c1[0]) == count[n]
        k = min(int(count[n]), 30)
        assert np.array_equal(a1[0, :k].cpu().numpy(), ans[n, :k])


# ------------------------------------------------------------------ BASELINE configs 4/5 shapes (fp32 path)
@pytest.mark.parametrize('arch_name,R,N', [('search-S', 448, 2), ('search-M', 512, 2), ('search-M', 448, 1),
                                           ('search-L', 512, 1)])
def test_native_resolutions_vs_oracle(arch_name, R, N):
    """28x28 / 32x32 / 56x56 / 112x112 planes, Cin > 32 blocks on planes > 16x16 (unfused path),
    deconv filters 64/40."""
    m, arch, sd = _model(arch_name)
    x = synth.make_images(N, R, seed=13)
    with torch.no_grad():
        ref = net_ref.forward(x, sd, arch)
    out = m(x.cuda())
    for a, b, name in zip(out, ref, ('out0', 'out1')):
        err = float(np.abs(a.cpu().numpy() - b.numpy()).max())
        print('%s@%d %s: max-abs err %.2e (|ref|max %.3f)' % (arch_name, R, name, err, float(b.abs().max())))
        assert err < OUT_ATOL, (arch_name, R, name, err)


def test_block_taps_tight_256():
    """Every block boundary of XS@256 against the oracle, tolerance scaled by the tap's magnitude."""
    m, arch, sd = _model('search-XS')
    x = synth.make_images(2, 256, seed=3)
    taps = {}
    with torch.no_grad():
        net_ref.forward(x, sd, arch, taps=taps)
    m(x.cuda())
    worst = (0.0, '')
    for name in ['first'] + ['stage.%d.%d' % (s, b) for s, nb in enumerate((6, 8, 10, 10)) for b in range(nb)] \
            + ['deconv.0', 'deconv.1', 'deconv.2']:
        ref = taps[name].numpy()
        got = m.tap(name).cpu().numpy().reshape(ref.shape)
        rel = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
        worst = max(worst, (rel, name))
        assert rel < 2e-5, (name, rel)
    print('worst scaled tap error %.2e at %s' % worst)


# ------------------------------------------------------------------ cross-stream isolation
def test_pipelined_halves_do_not_share_scratch():
    """infer_batch(pipeline_halves=True) runs two image halves on two streams; each half owns its TTA
    scratch.  Many small back-to-back batches (the second half's stage kernel is enqueued right after the
    first half's) must equal the single-stream result bitwise."""
    from litepose_amd import arch_zoo, config, engine
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=1.0)
    piped = engine.PoseEngine(cfg, arch, sd, person_capacity=30, pipeline_halves=True)
    plain = engine.PoseEngine(cfg, arch, sd, person_capacity=30, pipeline_halves=False)
    R = 64
    for it in range(24):
        N = 2 + 2 * (it % 3)
        x = synth.make_images(N, R, seed=300 + it).cuda()
        _, offs = _offsets(400 + it, N, R)
        a, c, s = [t.clone() for t in piped.infer_batch(x, offsets=offs)]
        dp, tp = [t.clone() for t in piped.last_maps()]
        b, d, u = plain.infer_batch(x, offsets=offs)
        dq, tq = plain.last_maps()
        assert torch.equal(dp, dq) and torch.equal(tp, tq), it
        assert torch.equal(c, d), it
        for n in range(N):
            k = min(int(c[n]), 30)
            assert torch.equal(a[n, :k], b[n, :k]) and torch.equal(s[n, :k], u[n, :k]), (it, n)


def test_detection_threshold_is_compared_in_float64():
    """group.py:38-41 compares float64(val) > 0.1: a peak of exactly float32(0.1) = 0.10000000149 passes
    (the device used to narrow the threshold to float32 and dropped it)."""
    from litepose_amd.core import group
    p = group.HeatmapParser(_cfg())
    ora = group_ref.HeatmapParser(group_ref.Params())
    H = W = 32
    det = np.zeros((1, 14, H, W), np.float32)
    tag = np.zeros((1, 14, H, W, 2), np.float32)
    for j in range(14):
        det[0, j, 5 + j, 7] = np.float32(0.1)          # exactly the float32 nearest to 0.1 (> 0.1 in float64)
        tag[0, j, 5 + j, 7] = 0.3
    det[0, 3, 20, 20] = np.nextafter(np.float32(0.1), np.float32(0))   # just below: rejected by both
    res = p.parse_batch(det, tag, True, False)
    a, s = ora.parse_image(det[0], tag[0], True, False)
    assert a.shape[0] == 1 and res[0][0].shape == a.shape
    assert np.array_equal(res[0][0], a) and np.array_equal(res[0][1], s)


# ------------------------------------------------------------------ N > 1 code path on the one GPU we have
def _bench_world2(tmp_path, extra, B, backend, world=2, shards=None):
    """`python bench.py --gpus N` with no launcher re-execs under torch.distributed.run; LP_BENCH_ONE_GPU=1 puts
    every rank on cuda:0.  The all-gathered records must be the single-rank runs (shards 0 .. N-1; `shards`: the ones
    compared) back to back, every rank must have replayed graphs, and the per-rank step times must be in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ['--steps', '2', '--warmup', '1', '--batch', str(B), '--no-cpu-baseline', '--no-kernel-profile',
              '--no-parity-check', '--no-io-leg', '--no-small-batch', '--long-steps', '0'] + extra
    env = dict(os.environ, LP_BENCH_BACKEND=backend, LP_BENCH_ONE_GPU='1')
    env.pop('RANK', None), env.pop('WORLD_SIZE', None), env.pop('LOCAL_RANK', None)
    g = str(tmp_path / 'gathered.npz')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--dump', g] + common,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    if r.returncode != 0 and backend == 'nccl':
        return None, r.stderr[-1500:]                 # RCCL refuses two ranks on one device on this box: the caller falls back
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == world and line['config']['global_batch'] == world * B and line['scaling'] == 'weak'
    assert line['graph_replay'] and len(line['per_rank']['ms_per_step']) == world, line['per_rank']
    assert line['graphs']['capture_failures'] == 0 and all(f == 0 for f in line['per_rank']['capture_failures'])
    gathered = np.load(g)
    assert gathered['count'].shape[0] == world * B
    for shard in (shards if shards is not None else range(world)):
        f = str(tmp_path / ('single%d.npz' % shard))
        r1 = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--shard-seed',
                             str(shard), '--dump', f] + common, env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, timeout=900)
        assert r1.returncode == 0, r1.stderr[-3000:]
        one = np.load(f)
        sl = slice(B * shard, B * shard + B)
        assert np.array_equal(gathered['count'][sl], one['count'])
        for n in range(B):
            k = min(int(one['count'][n]), 30)
            assert np.array_equal(gathered['kpts'][sl][n, :k], one['kpts'][n, :k])
            assert np.array_equal(gathered['scores'][sl][n, :k], one['scores'][n, :k])
    assert int(gathered['count'].sum()) >= world * B
    return line, ''


def test_bench_world2_gloo_on_one_gpu(tmp_path):
    """The N > 1 path of the headline configuration (gloo: works on any box)."""
    line, _ = _bench_world2(tmp_path, [], 8, 'gloo')
    assert line['config']['baseline_config'] is None


def test_bench_world2_config5_on_one_gpu(tmp_path):
    """BASELINE config 5 (search-M 512x512, bf16 storage, 32 images per GPU; here 4) through `--config 5` with two
    ranks on one GPU -- under RCCL ('nccl') if it accepts two ranks on one device, else gloo.  What the driver's
    8-GPU SCALE run of this workload must show is what is asserted here for two ranks: identical records to the
    single-rank runs of the same shards, every rank on graph replays, per-rank step times in the line
    (DESIGN.md section 5)."""
    line, why = _bench_world2(tmp_path, ['--config', '5'], 4, 'nccl')
    used = 'nccl'
    if line is None:
        print('two ranks on one device under nccl refused (%s); gloo instead' % why.strip().splitlines()[-1][:200])
        line, _ = _bench_world2(tmp_path, ['--config', '5'], 4, 'gloo')
        used = 'gloo'
    print('config 5, world 2 on one GPU, backend %s: per-rank ms/step %s' % (used, line['per_rank']['ms_per_step']))
    assert line['config']['baseline_config'] == 5 and line['dtype'] == 'bf16'
    assert 'LitePose-M@512' in line['metric']


def test_bench_world8_config5_gloo_on_one_gpu(tmp_path):
    """8-GPU readiness without 8 GPUs (the reference's only multi-GPU evaluation line is valid.py:165):
    `bench.py --gpus 8 --config 5` -- BASELINE config 5's launch shape -- with EIGHT ranks on the one GPU of the box (gloo):
    eight processes, eight engines, eight times the stage graphs captured next to each other's work, the affinity split
    eight ways, one all-gather of eight shards per step.  The gathered records must be the single-rank runs of the same
    shards (0, 3 and 7 are re-run and compared), every rank on graph replays with zero capture failures, every rank's
    step time and affinity record in the line.  No scaling number comes out of this (one GPU shared eight ways)."""
    line, _ = _bench_world2(tmp_path, ['--config', '5'], 2, 'gloo', world=8, shards=(0, 3, 7))
    assert line['config']['baseline_config'] == 5 and line['dtype'] == 'bf16' and line['n_gpus'] == 8
    aff = line['per_rank']['affinity']
    assert len(aff) == 8 and all(isinstance(a, dict) and 'pinned' in a for a in aff)
    pinned = [a for a in aff if a['pinned']]
    if pinned:                                           # sysfs readable: the node's cores were dealt to its eight ranks
        assert all(a['ranks_on_node'] == 8 for a in pinned)
        assert len({(a['first_core'], a['last_core']) for a in pinned}) == len(pinned)
    print('config 5, world 8 on one GPU (gloo): per-rank ms/step %s' % line['per_rank']['ms_per_step'])


def test_submit_graph_replay_equals_eager():
    """PoseEngine.submit captures a lane's batch into a hipGraph the second time it sees the same input
    buffers and replays it afterwards: records must stay bitwise those of the eager launches, also after the
    input buffers are re-filled in place with different images."""
    from litepose_amd import arch_zoo, config, engine
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
    N, R = 8, 256
    xs = [synth.make_images(N, R, seed=500 + k).cuda() for k in range(2)]
    offs_all = [_offsets(600 + k, N, R)[1] for k in range(2)]
    ref = []
    eng0 = engine.PoseEngine(cfg, arch, sd, person_capacity=30, graph=False)     # eager launches: the reference run
    for k in range(2):
        with eng0.submit(xs[k], offsets=offs_all[k]) as (a, c, s):
            ref.append((a.clone(), c.clone(), s.clone()))
    assert eng0.graph_stats()['graph_captures'] == 0 and not eng0._use_graphs
    eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
    xbuf = xs[0].clone()
    obuf = tuple(o.clone() for o in offs_all[0])
    seen_graph = False
    for it in range(16):                                 # buffer sets rotate: each captures on its 2nd batch, replays after
        k = (it // 3) % 2                                # content changes every three submits, buffers stay
        xbuf.copy_(xs[k])
        for dst, src in zip(obuf, offs_all[k]):
            dst.copy_(src)
        with eng.submit(xbuf, offsets=obuf) as (a, c, s):
            assert torch.equal(c, ref[k][1]), it
            for n in range(N):
                m = min(int(c[n]), 30)
                assert torch.equal(a[n, :m], ref[k][0][n, :m]) and torch.equal(s[n, :m], ref[k][2][n, :m]), (it, n)
        torch.cuda.synchronize()                         # the buffers are re-filled next iteration
        seen_graph = seen_graph or any(l['graphs'] for l in eng._lanes)
    assert seen_graph and eng._use_graphs, 'no lane captured a graph'


def test_submit_graphs_survive_a_shape_change_and_come_back():
    """a captured hipGraph holds raw pointers into the buffer set it was captured on.  Shape A
    until every set replays, then a partial batch (shape B: new buffers for every set), then A again -- the A graphs
    must either still own their buffers or not be replayed at all; with more than _MAX_SHAPES shapes in rotation the
    oldest is evicted and re-captured.  Every collected batch is compared with infer_batch."""
    from litepose_amd import arch_zoo, config, engine
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
    R, NA = 128, 8
    eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
    xa = synth.make_images(NA, R, seed=900).cuda()
    oa = _offsets(901, NA, R)[1]
    shapes = {NA: (xa, oa)}
    for nb in (6, 4, 2):                                   # partial batches: views of the full staging buffers
        shapes[nb] = (xa[:nb], tuple(torch.cat([o[:nb], o[NA:NA + nb]]).contiguous() for o in oa))
    ref = {}
    for nb, (xi, oi) in shapes.items():
        a, c, s = eng.infer_batch(xi.clone(), offsets=tuple(o.clone() for o in oi))
        ref[nb] = (a.clone(), c.clone(), s.clone())
    nset = eng.buffer_sets()

    def check(nb, tag):
        xi, oi = shapes[nb]
        with eng.submit(xi, offsets=oi) as (a, c, s):
            assert torch.equal(c, ref[nb][1]), (tag, nb)
            for n in range(nb):
                k = min(int(c[n]), 30)
                assert torch.equal(a[n, :k], ref[nb][0][n, :k]) and torch.equal(s[n, :k], ref[nb][2][n, :k]), (tag, nb, n)
        torch.cuda.synchronize()
    for it in range(3 * nset):                             # A: eager, capture, replay on every set
        check(NA, 'A%d' % it)
    r0 = eng.graph_stats()['graph_replays']
    assert r0 >= nset
    for it in range(nset):                                 # B once per set: new buffers next to A's
        check(6, 'B%d' % it)
    for it in range(2 * nset):                             # A again: must replay its OWN (still alive) buffers
        check(NA, 'A2_%d' % it)
    assert eng.graph_stats()['graph_replays'] >= r0 + 2 * nset
    for rnd in range(3):                                   # four shapes in rotation: evictions + re-captures
        for nb in (6, 4, 2, NA):
            for it in range(nset):
                check(nb, 'R%d_%d_%d' % (rnd, nb, it))
    st = eng.graph_stats()
    assert st['use_graphs'] and st['capture_failures'] == 0, st
    eng.reset_graphs()
    assert eng.graph_stats()['captured_sets'] == 0
    check(NA, 'after reset')


@pytest.mark.parametrize('launches,storage,iters', [('graph', 'f32', 5000), ('eager', 'f32', 3000),
                                                    ('graph', 'bf16', 3000)])
def test_serving_loop_has_no_wrong_batch_in_5000(launches, storage, iters):
    """Regression guard for the rare wrong batch: tools/flake_hunt.py -- the stress test's loop, every batch
    compared with a clean single-stream run.  Cause (round 4): packed fp32 instructions with op_sel:[0,1] in the small
    kernels of one network stream (dwpw_kernel's bias add, tta_project2x_kernel, refine_dm_kernel, ...) return a wrong
    low half in lanes 48-63 next to the bf16 MFMAs of the other stream's fused blocks; 2.5e-5 .. 2e-3 of the batches
    depending on which kernels could share a CU.  The library no longer contains the form (tests/test_host_cpu.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'tools', 'flake_hunt.py'), '--iters', str(iters), '--storage', storage]
    if launches == 'eager':
        cmd.append('--eager')
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith('iterations')][-1]
    print(last)
    assert 'mismatching batches 0 ' in last, r.stdout[-3000:]
    if launches == 'graph':
        assert "'capture_failures': 0" in last and "'use_graphs': True" in last, last


@pytest.mark.parametrize('mode', ['thread_local', 'global'])
def test_capture_with_a_second_thread_polling_events(mode):
    """De-risking the first multi-GPU run on one GPU: torch.distributed's RCCL watchdog is a second host thread that
    polls the events of OUTSTANDING collectives while this thread may be capturing a hipGraph.  tests/capture_probe.py
    (a subprocess: a broken capture can take the interpreter down) hammers event.query() / stream.query() from a
    second thread during prepare() and the first replays.  Required in either capture error mode: the process
    survives and every record is right -- an invalidated capture drops the engine to eager launches (visible in
    graph_stats and in bench.py's `graph_replay`), and those work right after the failed capture
    (lp_stream_abort_capture).  What the mode does with the foreign calls is printed; DESIGN 5b quotes it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'capture_probe.py'), mode],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if mode == 'global' and (r.returncode != 0 or not lines):
        # known on ROCm 7.2: under 'global' the foreign polls invalidate the capture AND leave the stream unusable
        # (launches keep failing with "previous error during capture" even after hipStreamEndCapture); that is why
        # 'thread_local' is the engine's default.  Reported, not required.
        pytest.xfail('capture_error_mode=global does not survive a polling thread: ' + r.stderr.strip()[-300:])
    assert r.returncode == 0 and lines, (r.returncode, r.stderr[-2000:])
    res = json.loads(lines[-1])
    if mode == 'thread_local':
        assert res['use_graphs'] and res['capture_failures'] == 0 and res['captured_sets'] == res['buffer_sets'], res
    print('capture under a polling thread:', res)
    assert res['records_ok'], res
    assert res['polls_ok'] + res['polls_raised'] > 0


@pytest.mark.parametrize('arch_name,H,W', [('search-XS', 256, 256), ('search-XS', 96, 160), ('search-L', 128, 128),
                                           ('search-XS', 80, 48)])
def test_fused_stem_vs_unfused_and_oracle(arch_name, H, W):
    """stem4_kernel (conv3x3 s2 + dw3x3 + 1x1 in one launch: conv and 1x1 on fp32 MFMAs, LDS tiles) against the unfused
    kernels (option "stem" = 0) on the stem tap, plain and mirrored (flip-TTA read), ragged tiles
    included, and the outputs against the oracle."""
    m, arch, sd = _model(arch_name)
    x = synth.make_images(3, H, seed=41, w=W).cuda()
    res = {}
    prev = m.get_option('stem')
    for mode in ('1', '0'):
        m.set_option('stem', int(mode))
        try:
            m.set_profiling(True)
            m.forward_native(x, flip=0)
            kernels = [n.split('|')[1] for n, _, _, _ in m.profile()]
            m.set_profiling(False)
            out = [o.clone() for o in m.forward_native(x, flip=2)]
            res[mode] = (out, m.tap('first').clone(), kernels)
        finally:
            m.set_option('stem', prev)
    assert 'stem4_kernel' in res['1'][2] and 'stem4_kernel' not in res['0'][2]
    a, b = res['1'][1], res['0'][1]
    rel = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
    print('%s %dx%d: fused vs unfused stem tap, scaled max diff %.2e, bitwise %s' % (arch_name, H, W, rel, torch.equal(a, b)))
    assert torch.equal(a, b), 'the fused stem sums in the order of the unfused kernels: bit-identical'
    for p_, q_ in zip(res['1'][0], res['0'][0]):
        assert torch.equal(p_, q_)
    with torch.no_grad():
        ref = net_ref.forward(x.cpu(), sd, arch)
        ref_f = net_ref.forward(torch.flip(x.cpu(), [3]), sd, arch)
    for k in range(2):
        np.testing.assert_allclose(res['1'][0][k][:3].cpu().numpy(), ref[k].numpy(), rtol=0, atol=OUT_ATOL)
        np.testing.assert_allclose(res['1'][0][k][3:].cpu().numpy(), ref_f[k].numpy(), rtol=0, atol=OUT_ATOL)


_DIAG_BODY = r'''
import ctypes as C, torch
from oracle import synth
from litepose_amd import _native as nv, arch_zoo, config
from litepose_amd.models import pose_mobilenet
arch = arch_zoo.get('search-XS')
m = pose_mobilenet.get_pose_net(config.get_cfg(), cfg_arch=arch)
m.load_state_dict(synth.make_state_dict(arch), strict=True)
x = synth.make_images(4, 256, seed=43).cuda()
ref = [o.clone() for o in m.forward_native(x, flip=2)]
nv.lib().lp_diag_read(None, 0, 1)
m.set_option('stem', 0)
m.set_option('diag_dwpw', 1)
m.set_profiling(True)
for _ in range(20):
    out = m.forward_native(x, flip=2)
kernels = [n.split('|')[1] for n, _, _, _ in m.profile()]
m.set_profiling(False)
assert 'dwpw_kernel' in kernels
for p_, q_ in zip(out, ref):
    assert torch.equal(p_, q_)
torch.cuda.synchronize()
buf = (C.c_uint32 * 17)()
assert nv.lib().lp_diag_read(C.cast(buf, C.c_void_p), 17, 1) == 0
m.set_option('diag_dwpw', 2)
m.forward_native(x, flip=2)
torch.cuda.synchronize()
assert nv.lib().lp_diag_read(C.cast(buf, C.c_void_p), 17, 1) in (1, 2)   # one per launch (plain / mirrored half)
r = list(buf)[1:]
assert r[0] == 5 and r[1] == 1 and r[2] == (11 | (1 << 8))          # workgroup 5, wave 1, dword 2*4+3, epilogue
assert (r[3] | (r[4] << 32)) == 1 << 37 and (r[5] | (r[6] << 32)) == 0   # lane 37; the re-fetch is right
assert r[7] ^ r[8] == 0x00010000
print('DIAG OK')
'''


def test_diagnostic_variants_are_bit_identical_and_log_nothing_on_a_quiet_gpu():
    """DESIGN 5b diagnostics (tools/flake_hunt.py --diag): the self-checking dwpw_kernel<3, ..., DIAG> (option
    "diag_dwpw", with "stem" = 0) computes what the shipped kernels compute, bit for bit; with ONE network in flight its
    bias registers never disagree with their scalar-cache copy (round 3's wrong batches took two networks in flight); the
    positive control ("diag_dwpw" = 2: one flipped bit in one lane per launch) is logged as an epilogue event with the
    lane, the register and the two values.  Round 5: that variant keeps the erratum-prone packed form on purpose, so it is
    linked into the diagnostics flavour only (lib/liblitepose_amd_diag.so, LP_NATIVE_FLAVOUR=diag, own process); the
    product library must refuse the option and the log read."""
    import subprocess
    import sys
    from litepose_amd import _native as nv
    m, arch, sd = _model('search-XS')
    with pytest.raises(Exception, match='diagnostics'):
        m.set_option('diag_dwpw', 1)
    assert m.get_option('diag_dwpw') == 0
    assert nv.lib().lp_diag_read(None, 0, 0) == -8                           # LP_ERR_UNSUPPORTED
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from litepose_amd import build as _b
    _b.build(flavour='diag', verbose=False)          # up to date after __graft_entry__.build(); never a silent skip
    r = subprocess.run([sys.executable, '-c', _DIAG_BODY], cwd=root, env=dict(os.environ, LP_NATIVE_FLAVOUR='diag'),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and 'DIAG OK' in r.stdout, r.stdout[-3000:]


def test_regstage_flavour_is_bit_identical():
    """lib/liblitepose_amd_regstage.so (python -m litepose_amd.build --flavour regstage: the fused block kernels stage their
    weights through registers instead of LDS-DMA and claim whole CUs; the A/B of DESIGN 5b) computes the same bits as the
    library."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from litepose_amd import build as _b
    _b.build(flavour='regstage', verbose=False)      # up to date after __graft_entry__.build(); never a silent skip
    code = ("import torch, hashlib; from oracle import synth; from litepose_amd import arch_zoo, config; "
            "from litepose_amd.models import pose_mobilenet; arch = arch_zoo.get('search-XS'); "
            "m = pose_mobilenet.get_pose_net(config.get_cfg(), cfg_arch=arch); "
            "m.load_state_dict(synth.make_state_dict(arch), strict=True); "
            "o = m.forward_native(synth.make_images(4, 256, seed=43).cuda(), 2); "
            "print('SUM', [hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest() for t in o])")
    outs = []
    for fl in ('', 'regstage'):
        env = dict(os.environ, LP_NATIVE_FLAVOUR=fl)
        if not fl:
            env.pop('LP_NATIVE_FLAVOUR')
        r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith('SUM')][-1])
    assert outs[0] == outs[1]


def test_submit_split_schedule_stress_two_inputs_in_flight():
    """PoseEngine.submit keeps pipeline_depth + 1 batches in flight on two NET streams + one AE stream with four
    buffer sets.  A serving loop with one staging buffer per buffer set (re-filled in place, so every set replays
    its hipGraphs) alternates two different inputs without any host synchronisation; every collected batch must
    equal the un-pipelined result of its input, read right after result() (a buffer set re-used too early, an AE
    stage reading the maps of the wrong batch, or a graph node running out of order shows up as a mismatch: the
    hipMemsetAsync node that used to clear the records did exactly that with two AE graphs back to back)."""
    from litepose_amd import arch_zoo, config, engine, parallel
    arch = arch_zoo.get('search-XS')
    cfg = config.apply_arch(_cfg(), arch)
    sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
    N, R = 8, 256
    xs = [synth.make_images(N, R, seed=700 + k).cuda() for k in range(2)]
    offs_all = [_offsets(800 + k, N, R)[1] for k in range(2)]
    eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
    ref = []
    for k in range(2):
        a, c, s = eng.infer_batch(xs[k], offsets=offs_all[k])
        ref.append((a.clone(), c.clone(), s.clone()))
    assert sum(int(r[1].sum()) for r in ref) >= 8
    depth = eng.pipeline_depth()
    nset = depth + 2
    stage = [(xs[0].clone(), tuple(o.clone() for o in offs_all[0])) for _ in range(nset)]
    pend, bad = [], []

    def collect():
        k, h = pend.pop(0)
        a, c, s = h.result()
        # what the all-gather does: allocate + pack right behind result() (new allocations must not alias
        # anything a graph still writes)
        ka, ca, sa = parallel.unpack_records(parallel.pack_records(a, c, s).clone(), 30, a.shape[2], a.shape[3])
        h.release()
        if not (torch.equal(ca, ref[k][1]) and all(
                torch.equal(ka[n, :min(int(ca[n]), 30)], ref[k][0][n, :min(int(ca[n]), 30)]) and
                torch.equal(sa[n, :min(int(ca[n]), 30)], ref[k][2][n, :min(int(ca[n]), 30)]) for n in range(N))):
            bad.append((len(bad), k))
    for it in range(48):
        k = (it // 3 + it) % 2                            # irregular alternation: every set sees both inputs
        xb, ob = stage[it % nset]                         # its previous batch was collected: safe to re-fill
        xb.copy_(xs[k])
        for dst, src in zip(ob, offs_all[k]):
            dst.copy_(src)
        pend.append((k, eng.submit(xb, offsets=ob)))
        if len(pend) > depth:
            collect()
    while pend:
        collect()
    if bad:
        # tell a wrong REFERENCE (infer_batch on a fresh engine) from wrong pipelined batches: recompute it
        again = []
        for k in range(2):
            a, c, s = eng.infer_batch(xs[k], offsets=offs_all[k])
            again.append(bool(torch.equal(c, ref[k][1]) and torch.equal(a, ref[k][0]) and torch.equal(s, ref[k][2])))
        raise AssertionError('mismatching batches %s of 48 (by input: %s); reference reproducible: %s'
                             % ([b[1] for b in bad], {k: sum(1 for b in bad if b[1] == k) for k in (0, 1)}, again))
    assert eng._use_graphs and all(ln['graphs'] for ln in eng._lanes), 'the sets did not replay graphs'
    st = eng.graph_stats()
    assert st['graph_replays'] >= 48 - 2 * nset and st['capture_failures'] == 0, st


def test_packed_fp32_op_sel_erratum_reproducer_controls_are_clean():
    """tools/ubench/pk_vs_mfma.hip: register-only victim waves next to MFMA aggressor waves.  Asserted: the
    CONTROLS are clean -- no aggressor, plain v_add_f32, the packed forms without op_sel -- i.e. the checker itself is
    sound on this box.  Reported, not asserted (it is the hardware's behaviour, and a fixed part would be good news): the
    rate of wrong results of `v_pk_add_f32 ... op_sel:[0,1]` next to bf16 MFMAs, which is what the library must not
    contain (tests/test_host_cpu.py scans the build for it)."""
    import re
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, 'tools', 'ubench', 'bin', 'pk_vs_mfma')
    tmp = None
    if not os.path.exists(exe):
        hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
        if not os.path.exists(hipcc):
            pytest.skip('reproducer binary not built and no hipcc')
        tmp = tempfile.mkdtemp()
        exe = os.path.join(tmp, 'pk_vs_mfma')
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-o', exe,
                            os.path.join(root, 'tools', 'ubench', 'pk_vs_mfma.hip')], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300)
        if r.returncode != 0:
            pytest.skip('reproducer did not compile: ' + r.stdout[-300:])
    try:
        r = subprocess.run([exe, '0.3'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)
    assert r.returncode == 0, r.stdout[-1000:]
    rows = []
    for ln in r.stdout.splitlines():
        m = re.match(r'cfg\s+(\d+)\s+aggressor (.+?)\s+victim (.+?)\s+[\d.]+ s .*wrong lane-results (\d+) =', ln)
        if m:
            rows.append((m.group(2).strip(), m.group(3).strip(), int(m.group(4))))
    assert len(rows) >= 20, r.stdout[-1500:]
    hits = [(a, v, n) for a, v, n in rows if n]
    print('configurations with wrong results:', hits)
    for a, v, n in rows:
        control = a == 'none' or v.startswith('v_add_f32') or v in ('v_pk_add_f32', 'v_pk_fma_f32') or 'op_sel:' not in v
        if control:
            assert n == 0, (a, v, n)
