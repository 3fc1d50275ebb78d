"""Host-side logic and the C-ABI surface, no GPU needed."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import spec


def test_library_exports_every_declared_symbol():
    from litepose_amd import _native as nv
    hdr = open(os.path.join(ROOT, 'include', 'litepose_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(lp_[a-z_0-9]+)\s*\(', hdr))
    assert len(declared) >= 20
    lib = nv.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), 'missing export ' + name
    assert declared == set(nv.EXPORTS), declared ^ set(nv.EXPORTS)
    assert b'gfx950' in lib.lp_version()


def test_arch_zoo_and_key_scheme_match_oracle_spec():
    from litepose_amd import arch_zoo, config
    from litepose_amd.models import pose_mobilenet
    for name in arch_zoo.names():
        arch = arch_zoo.get(name)
        m = pose_mobilenet.get_pose_net(config.get_cfg(), cfg_arch=arch)
        keys = m.keys()
        ref = spec.state_dict_shapes(arch)
        assert [k for k, _ in keys] == list(ref.keys()), name
        for k, shp in keys:
            assert tuple(shp) == tuple(ref[k]), (name, k)
    arch = arch_zoo.get('search-XS')
    m = pose_mobilenet.get_pose_net(config.get_cfg('coco'), cfg_arch=arch)
    assert m.final_channel == [34, 17]


def test_native_error_reporting_without_gpu():
    from litepose_amd import _native as nv, arch_zoo, config
    from litepose_amd.models import pose_mobilenet
    lib = nv.lib()
    m = pose_mobilenet.get_pose_net(config.get_cfg(), cfg_arch=arch_zoo.get('search-XS'))
    shp = (C.c_int64 * 4)(1, 2, 3, 4)
    buf = (C.c_float * 24)()
    assert lib.lp_net_set_weight(m._h, b'no.such.key', buf, shp, 4) == -2
    assert b'no.such.key' in lib.lp_last_error()
    assert lib.lp_net_set_weight(m._h, b'first.0.0.weight', buf, shp, 4) == -3
    assert lib.lp_net_workspace_bytes(m._h, 1, 64, 64) == 0           # not finalized
    with pytest.raises(nv.LitePoseNativeError):
        m.forward_native(torch.zeros(1, 3, 64, 64))


def test_config_merge_and_arch_override(tmp_path):
    from litepose_amd import arch_zoo, config
    cfg = config.get_cfg()
    y = tmp_path / 'mobile.yaml'
    y.write_text('DATASET:\n  NUM_JOINTS: 14\n  MAX_NUM_PEOPLE: 30\nTEST:\n  FLIP_TEST: False\n  SCALE_FACTOR: [1]\n')
    cfg.merge_from_file(str(y))
    cfg.merge_from_list(['TEST.DETECTION_THRESHOLD', '0.2'])
    assert cfg.TEST.FLIP_TEST is False and cfg.TEST.DETECTION_THRESHOLD == 0.2
    config.apply_arch(cfg, arch_zoo.get('search-S'))
    assert cfg.DATASET.INPUT_SIZE == 448 and cfg.DATASET.OUTPUT_SIZE == [112, 224]


def test_parser_params_and_capacity():
    from litepose_amd import config
    from litepose_amd.core import group
    p = group.HeatmapParser(config.get_cfg())
    assert p.person_capacity == 14 * 30
    assert list(p._q.joint_order[:14]) == [0, 1, 2, 3, 4, 5, 6, 11, 12, 7, 8, 9, 10, 13]
    cfg = config.get_cfg()
    cfg.TEST.DETECTION_THRESHOLD = -0.5
    with pytest.raises(ValueError):
        group.HeatmapParser(cfg)


def test_multi_scale_size_matches_oracle():
    from litepose_amd.utils import transforms
    from oracle import transforms_ref
    for hw in ((480, 640), (640, 480), (256, 256), (333, 500)):
        a = transforms.get_multi_scale_size(hw, 256, 1.0, 1.0)
        b = transforms_ref.get_multi_scale_size(hw, 256, 1.0, 1.0)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.allclose(a[2], b[2])


def test_shard_and_record_packing():
    from litepose_amd import parallel
    for n, w in ((64, 8), (10, 4), (3, 8)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    k = torch.randn(5, 30, 14, 5)
    c = torch.tensor([0, 3, 30, 41, 7], dtype=torch.int32)
    s = torch.randn(5, 30)
    k2, c2, s2 = parallel.unpack_records(parallel.pack_records(k, c, s), 30, 14, 5)
    assert torch.equal(k, k2) and torch.equal(c, c2) and torch.equal(s, s2)


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from litepose_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
total = 6
g = torch.Generator().manual_seed(0)
K = torch.randn(total, 30, 14, 5, generator=g); S = torch.randn(total, 30, generator=g)
Cn = torch.arange(total, dtype=torch.int32) * 7
a, b = parallel.shard_range(total, rank, world)
k, c, s = parallel.all_gather_records(K[a:b].contiguous(), Cn[a:b].contiguous(), S[a:b].contiguous())
assert torch.equal(k, K) and torch.equal(c, Cn) and torch.equal(s, S)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_all_gather_records_gloo_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_result_writer_matches_reference_json(tmp_path):
    """records -> evaluator JSON: identical (after json parsing AND as text) to what the real
    CrowdPoseDataset.evaluate wrote for the same predictions (tests/golden/gen_golden_results.py)."""
    import json
    from litepose_amd import results
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'results_inputs.npz'))
    want_txt = open(os.path.join(ROOT, 'tests', 'golden', 'results_golden.json')).read()
    res = results.records_to_results(g['kpts'], g['count'], g['scores'], g['ids'])
    assert res == json.loads(want_txt)
    f = results.write_results(res, str(tmp_path / 'r.json'))
    assert open(f).read() == want_txt
    # the valid.py accumulators give the same list
    preds = [[g['kpts'][n, p] for p in range(int(g['count'][n]))] for n in range(len(g['ids']))]
    scs = [[float(g['scores'][n, p]) for p in range(int(g['count'][n]))] for n in range(len(g['ids']))]
    assert results.preds_to_results(preds, scs, g['ids']) == res
    assert abs(results.person_area(preds[0][0]) - float(
        (preds[0][0][:, 0].max() - preds[0][0][:, 0].min()) * (preds[0][0][:, 1].max() - preds[0][0][:, 1].min()))) == 0


def test_enable_center_and_used_joints():
    """default.py:173-177 semantics: WITH_CENTER adds one joint to every stage; IGNORE_CENTER removes it
    again from the merged maps / records (inference.py:148-150, group.py:110-111)."""
    from litepose_amd import config
    from litepose_amd.core import inference
    cfg = config.get_cfg('crowd_pose')
    assert inference.used_joints(cfg) == 14
    config.enable_center(cfg)
    assert cfg.DATASET.WITH_CENTER and cfg.DATASET.NUM_JOINTS == 15 and cfg.MODEL.NUM_JOINTS == 15
    assert inference.used_joints(cfg) == 14
    assert inference.flip_index_for(cfg)[-1] == 14          # the centre joint maps to itself
    config.enable_center(cfg)                               # idempotent
    assert cfg.DATASET.NUM_JOINTS == 15
    coco = config.enable_center(config.get_cfg('coco'), ignore_center=False)
    assert coco.DATASET.NUM_JOINTS == 18 and inference.used_joints(coco) == 18


def test_result_writer_edge_cases():
    from litepose_amd import results
    k = np.zeros((2, 3, 14, 5), np.float32)
    assert results.records_to_results(k, np.array([0, 0]), np.zeros((2, 3), np.float32), [7, 8]) == []
    with pytest.raises(ValueError):
        results.records_to_results(k, np.array([4, 0]), np.zeros((2, 3), np.float32), [7, 8])
    with pytest.raises(ValueError):
        results.records_to_results(k, np.array([0, 0]), np.zeros((2, 3), np.float32), [7])
    k[1, 0, :, 0] = np.arange(14)
    k[1, 0, :, 1] = 2 * np.arange(14)
    r = results.records_to_results(k, np.array([0, 1]), np.ones((2, 3), np.float32), [7, 8], num_joints=13)
    assert len(r) == 1 and r[0]['image_id'] == 8 and len(r[0]['keypoints']) == 39
    assert r[0]['bbox'] == [0.0, 0.0, 12.0, 24.0]


def test_update_config_mirrors_reference_postprocessing(tmp_path):
    """lib/config/default.py:156-186: YAML + opts merge, WITH_CENTER -> NUM_JOINTS += 1 (both nodes),
    scalar OUTPUT_SIZE / loss switches become lists."""
    import types
    from litepose_amd import config
    y = tmp_path / 'exp.yaml'
    y.write_text('DATASET:\n  WITH_CENTER: true\n  OUTPUT_SIZE: 64\n  NUM_JOINTS: 14\n'
                 'LOSS:\n  WITH_AE_LOSS: true\nTEST:\n  DETECTION_THRESHOLD: 0.2\n')
    args = types.SimpleNamespace(cfg=str(y), opts=['TEST.FLIP_TEST', 'False', 'TEST.NMS_KERNEL', '3'])
    cfg = config.update_config(config.get_cfg('crowd_pose'), args)
    assert cfg.DATASET.NUM_JOINTS == 15 and cfg.MODEL.NUM_JOINTS == 15 and cfg.DATASET.WITH_CENTER
    assert cfg.DATASET.OUTPUT_SIZE == [64] and cfg.LOSS.WITH_AE_LOSS == [True]
    assert cfg.TEST.FLIP_TEST is False and cfg.TEST.NMS_KERNEL == 3 and cfg.TEST.DETECTION_THRESHOLD == 0.2
    plain = config.update_config(config.get_cfg('coco'), None)
    assert plain.DATASET.NUM_JOINTS == 17 and not plain.DATASET.WITH_CENTER


def test_project2image_rules_of_aggregate_results():
    """lib/core/inference.py:176-208 on the reference-shaped API.  TEST.PROJECT2IMAGE = True: the maps of every scale come
    projected to one size -- a mismatch is an error, never a sum of mismatched planes.  TEST.PROJECT2IMAGE = False (built in
    round 6: :180-189, :201-206): the first scale's maps define the size, later scales are RESIZED on the device (needs a GPU:
    a CPU tensor fails loudly at the native call; values are pinned by tests/test_gpu_parity.py against golden_ms nop2i*)."""
    import torch
    from litepose_amd import _native, config
    from litepose_amd.core import inference
    cfg = config.get_cfg('crowd_pose')
    cfg.TEST.SCALE_FACTOR = [2, 1]
    big = inference._Merged([torch.zeros(1, 14, 128, 128)])
    big_t = inference._Merged([torch.zeros(1, 14, 128, 128, 2)])
    small = inference._Merged([torch.zeros(1, 14, 64, 64)])
    small_t = inference._Merged([torch.zeros(1, 14, 64, 64, 2)])
    for p2i in (True, False):
        cfg.TEST.PROJECT2IMAGE = p2i
        final, tags = inference.aggregate_results(cfg, 2, None, [], big, big_t)      # first scale: accepted as it comes
        assert tags == [] and final.shape == (1, 14, 128, 128)
        # a second scale of another size: an error with PROJECT2IMAGE (the merge should have projected it), a device resize
        # without (CPU tensors here: the native layer refuses them -- there is no CPU fallback)
        with pytest.raises(ValueError if p2i else _native.LitePoseNativeError):
            inference.aggregate_results(cfg, 1, final, tags, small, small_t)
    with pytest.raises(TypeError):
        inference.aggregate_results(cfg, 1, None, [], [torch.zeros(1)], [torch.zeros(1)])


def _kernel_resources():
    import json
    from litepose_amd import build as _b
    path = os.path.join(_b.LIBDIR, 'kernel_resources.json')
    if not os.path.exists(path):
        pytest.skip('no kernel resource report (written by `python -m litepose_amd.build`)')
    res = json.load(open(path))
    assert len(res) > 100, 'resource report looks empty'
    return res


def test_no_kernel_uses_scratch():
    """No kernel of the build spills to scratch: a spill inside a depthwise loop costs more than the fusion saves, and
    lp::uses_scratch() (asked by every fused-block launcher) would silently route such a variant to the unfused chain.
    (Round 3 believed spilling kernels caused wrong batches; round 4 found the cause elsewhere)  Checked on
    the resource report of the build (hipcc -Rpass-analysis=kernel-resource-usage -> lib/kernel_resources.json)."""
    res = _kernel_resources()
    bad = {k: v['scratch'] for k, v in res.items() if v.get('scratch', 0) > 0}
    assert not bad, bad
    # what the default path of the headline configuration launches must be there at all
    for k in ('lp::mb16_kernel<5, 3, true>', 'lp::mb16_kernel<3, 2, true>', 'lp::mb16_kernel<3, 3, false>',
              'lp::mbt_kernel<2, 1, true>', 'lp::mbt_s2_kernel<1, 1>', 'lp::mbconv2_kernel<true, 8, 1>',
              # bf16 storage, S@448 / M@512 (BASELINE configs 4 / 5): the fused blocks of every stage
              'lp::mbtb_kernel<1, 1, true>', 'lp::mbtb_kernel<2, 1, true>', 'lp::mbtb_kernel<3, 2, true>',
              'lp::mbtb_kernel<3, 4, false>', 'lp::mbtb_kernel<5, 3, true>', 'lp::mbtb_kernel<5, 4, false>',
              'lp::mbtb_kernel<8, 4, true>'):
        assert k in res, k


def test_library_has_no_packed_fp32_with_op_sel_01():
    """DESIGN 5b, what round 4's hunt for the rare wrong batch ended in: on gfx950 a packed fp32 instruction
    (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose op_sel takes src0's LOW and src1's HIGH register for the low result
    returns src0.lo (+|*) 0 in lanes 48-63 while waves that issue bf16 MFMAs run next to it
    (tools/ubench/pk_vs_mfma.hip reproduces it in seconds).  hipcc builds such forms from ordinary float2 arithmetic and
    from broadcasting a wave-uniform scalar that sits in an odd SGPR.  The library therefore (a) compiles the files
    without hand-placed packed FMAs without packed fp32 (build.py NOPK), (b) feeds the packed FMAs of the unfused
    depthwise kernels aligned (w, w) pairs (engine.cpp pack_dw_dup) -- and the build (build.py _scan) as well as this test
    disassemble the result.  Round 5: the rule is an ALLOWLIST -- every packed fp32 instruction's modifier form must be one
    the reproducer ran clean (scan_isa.CLEAN); the known-bad routing and any form nobody has tested fail alike -- the
    product library has NO exception (the self-checking dwpw variant that keeps the bad form lives in the diagnostics
    flavour only), and the scan must leave the library's bytes alone (round 4's objcopy call rewrote it in place)."""
    import hashlib
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'litepose_amd', 'lib', 'liblitepose_amd.so')
    if not os.path.exists(lib):
        pytest.skip('library not built')
    spec = importlib.util.spec_from_file_location('scan_isa', os.path.join(root, 'tools', 'scan_isa.py'))
    si = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(si)
    if not si.tools_present():
        pytest.skip('ROCm LLVM tools not found')
    sha = lambda p: hashlib.sha256(open(p, 'rb').read()).hexdigest()     # noqa: E731
    before = sha(lib)
    r = si.scan(lib)
    assert sha(lib) == before, 'the scan edited the library it certifies'
    assert r['kernels'] > 150 and r['pk_total'] > 5000, 'disassembly looks empty: the detector would see nothing'
    assert r['pk_op_sel_01'] == {}, r['pk_op_sel_01']
    assert r['pk_unverified'] == {}, (r['pk_unverified'], r['forms'])
    # the detector detects: the known-bad routings are not on the allowlist, the forms the library uses are
    assert not (si.KNOWN_BAD & si.CLEAN)
    assert all(any(f.startswith(m) for m in ('v_pk_add_f32', 'v_pk_mul_f32', 'v_pk_fma_f32')) for f in r['forms'])
    # LDS-DMA: only the fused block kernels stage weights that way (cleared by the regstage A/B, kernels.h)
    assert r['lds_dma'] and all(k.startswith(('lp::mb16_kernel<', 'lp::mbt_kernel<', 'lp::mbt_s2_kernel<', 'lp::mbtb_kernel<',
                                               'lp::mbtb_s2_kernel<', 'lp::mbtq_kernel<', 'lp::mbtd_kernel<')) for k in r['lds_dma']), r['lds_dma']
    alt = os.path.join(root, 'litepose_amd', 'lib', 'liblitepose_amd_regstage.so')
    if os.path.exists(alt):
        assert si.scan(alt)['lds_dma'] == {}
    diag = os.path.join(root, 'litepose_amd', 'lib', 'liblitepose_amd_diag.so')
    if os.path.exists(diag):       # positive control on real compiler output: the diagnostic variant keeps the bad form
        rd = si.scan(diag)
        assert rd['pk_op_sel_01'] and all(k.startswith('lp::dwpw_kernel<') and k.endswith(', true>') for k in rd['pk_unverified'])


def test_lds_layouts_of_the_fused_blocks_in_the_bank_model():
    """tools/lds_model.py restates the LDS addresses of the fused InvBottleneck kernels lane by lane and counts LDS
    cycles with the instruction-specific lane groups and banks of MI355X_MICROARCH.md.  Pinned here: the depthwise row
    reads of every kernel are conflict-free as ds_read_b128 (what keep_b128 preserves), the ds_read2_b64 pair hipcc
    built from the half-used slots was not (the 31 % of round 2's PMC pass), and the depthwise-result buffer of the bf16
    kernels needs its row swap."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('lds_model', os.path.join(root, 'tools', 'lds_model.py'))
    lm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lm)
    m = lm.mb16()
    assert m['depthwise rows as 48 ds_read_b128'] == (192, 192)
    got, free = m['the outer half-slots as 8 ds_read2_b64 (what hipcc built)']
    assert got - free == 192                                        # of the 240 conflict cycles per chunk PMC counted
    assert lm.mbt()['depthwise rows as 48 ds_read_b128'] == (192, 192)
    s2 = lm.mbt_s2()
    assert s2['depthwise rows (even / odd planes), 45 ds_read_b128'] == (180, 180)
    assert s2['depthwise result, 2 ds_write_b128'] == (16, 16) and s2['project operands, 4 ds_read_b64'] == (8, 8)
    b = lm.mbtb()
    assert b['depthwise result, 2 ds_write_b128, rows in order'] == (32, 16)
    assert b['depthwise result, 2 ds_write_b128, rows swapped on odd row pairs'] == (16, 16)
    assert b['project operands of the 8 waves, 32 ds_read_b32'] == (64, 64)


def test_isa_scan_classifier_on_the_reproducers_rows():
    """tools/scan_isa.classify on hand-written disassembly lines: the routings tools/ubench/pk_vs_mfma.hip counted wrong
    results for are 'known_bad', the forms it ran clean are 'clean', and a form nobody has run (or any form with neg_lo /
    neg_hi, which the library does not use) is 'untested' -- both of the latter two fail the build (build.py _scan).  Needs no
    GPU, no library and no ROCm tools: the rule itself is pinned here."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('scan_isa', os.path.join(root, 'tools', 'scan_isa.py'))
    si = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(si)
    v = lambda line: si.classify(line)[3]                          # noqa: E731
    assert si.classify('\tv_add_f32_e32 v1, v2, v3') is None
    assert si.classify('\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]') is None     # not arithmetic (ran clean anyway)
    # clean rows of profiles/r05_pk_vs_mfma.txt
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]   // 0123') == 'clean'
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]') == 'clean'
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]') == 'clean'
    assert v('\tv_pk_mul_f32 v[0:1], v[2:3], s[4:5] op_sel_hi:[1,0]') == 'clean'
    assert v('\tv_pk_fma_f32 v[0:1], v[2:3], s[8:9], v[0:1]') == 'clean'
    assert v('\tv_pk_fma_f32 v[0:1], v[2:3], s[8:9], v[0:1] op_sel_hi:[1,0,1]') == 'clean'
    assert v('\tv_pk_fma_f32 v[0:1], v[2:3], v[8:9], v[0:1] op_sel_hi:[1,1,0]') == 'clean'
    # the erratum's routings
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]') == 'known_bad'
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]') == 'known_bad'
    assert v('\tv_pk_mul_f32 v[0:1], v[2:3], s[4:5] op_sel:[0,1]') == 'known_bad'
    assert v('\tv_pk_fma_f32 v[0:1], v[2:3], s[8:9], v[0:1] op_sel:[0,1,0] op_sel_hi:[1,1,0]') == 'known_bad'
    # ran clean in the reproducer but not used by the library: deliberately NOT on the allowlist (the list stays as small
    # as the library's needs) -- and a routing nobody has run
    assert v('\tv_pk_fma_f32 v[0:1], v[2:3], v[8:9], v[0:1] op_sel:[0,0,1]') == 'untested'
    assert v('\tv_pk_fma_f32 v[0:1], v[2:3], v[8:9], v[0:1] op_sel:[1,1,1] op_sel_hi:[0,0,0]') == 'untested'
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]') == 'untested'
    assert v('\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]') != 'clean'


def test_ae_path_selection_rules(monkeypatch):
    """PoseEngine._ae_path (no GPU needed: the rule reads cfg, the parser parameters and the engine OPTIONS only).  'mid' --
    nothing materialised -- is the default wherever its walk kernels apply (NMS_KERNEL 3 / 5); NMS_KERNEL 7 keeps rounds
    2-4's 'dm'; shapes outside the exact x2 projection take the reference-shaped 'maps'.  Round 6: the path is a constructor
    option (ae='mid' | 'dm' | 'maps'); environment variables do not reach the engine (only through
    engine.options_from_env, which bench.py / tools call)."""
    import types
    from litepose_amd import config, engine
    monkeypatch.setenv('LP_AE', 'maps')                        # must NOT be seen by the engine
    monkeypatch.setenv('LP_AE_MID', '0')

    def path(H, W, nms=5, project=True, tpj=True, people=30, **opt):
        cfg = config.get_cfg()
        cfg.TEST.NMS_KERNEL, cfg.TEST.NMS_PADDING = nms, nms // 2
        cfg.TEST.PROJECT2IMAGE = project
        cfg.MODEL.TAG_PER_JOINT = tpj
        stub = types.SimpleNamespace(cfg=cfg, parser=types.SimpleNamespace(params=types.SimpleNamespace(max_num_people=people)),
                                     options=engine.make_options(**opt))
        return engine.PoseEngine._ae_path(stub, H, W)

    assert path(256, 256) == 'mid' and path(448, 448) == 'mid' and path(512, 512) == 'mid' and path(256, 256, nms=3) == 'mid'
    assert path(256, 256, nms=7) == 'dm'                       # radius 3: the walk kernel is not instantiated
    assert path(1024, 1024) == 'mid' and path(256, 1028) == 'maps' and path(254, 254) == 'maps'   # W % 4, W <= 1024
    assert path(256, 256, project=False) == 'maps' and path(256, 256, tpj=False) == 'maps' and path(256, 256, people=65) == 'maps'
    assert path(256, 256, ae='dm') == 'dm' and path(256, 256, ae='maps') == 'maps'
    assert path(256, 256, nms=7, ae='mid') == 'mid'            # forced: lp_parse_mid falls back to its band kernel
    assert path(254, 254, ae='mid') == 'maps'                  # an option never overrides the kernels' gates
    with pytest.raises(ValueError):
        engine.make_options(ae='nonsense')
    with pytest.raises(ValueError):
        engine.make_options(shed='split')
    # the env translation used by bench.py / tools: explicit, complete, and LP_AE_MID is gone
    o = engine.options_from_env({'LP_AE': 'dm', 'LP_GRAPH': '0', 'LP_NET_PRIO': '0', 'LP_LANES': '6', 'LP_AE_MID': '1'})
    assert o == {'ae': 'dm', 'graph': False, 'net_prio': 0, 'lanes': 6}
    assert engine.make_options(o)['sched'] == 'split' and engine.options_from_env({}) == {}
