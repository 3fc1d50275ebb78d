"""bench.py's own arithmetic, without a GPU: the roofline denominators and the lookup of the
committed PMC traffic.  The byte counts pinned here are the ones every committed bench line divides by."""
import importlib.util
import json
import os

import pytest

from conftest import load_arch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('lp_bench', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_path_bytes_of_the_headline_config(bench):
    """N*(F*B_op + B_post) of XS@256 b64 with flip-TTA: the bytes_per_step of every committed headline line."""
    arch = load_arch('search-XS')
    b_op, b_post = bench.algorithmic_bytes_per_image(arch, 14, 256, True)
    assert 64 * (2 * b_op + b_post) == 18259902464
    # B_post = the two stage outputs of image + mirror: 2 * 4 B * (28 * 64^2 + 14 * 128^2)
    assert b_post == 2 * 4 * (28 * 64 * 64 + 14 * 128 * 128)
    line = json.loads(open(os.path.join(ROOT, 'profiles', 'r02_bench_n1.json')).read().strip().splitlines()[-1])
    assert line['path_roofline']['bytes_per_step'] == 18259902464
    assert abs(line['path_roofline']['frac'] -
               18259902464 / (line['ms_per_step'] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS) < 1e-3


def test_bf16_storage_halves_the_stored_activations_only(bench):
    arch = load_arch('search-S')
    f32, post32 = bench.algorithmic_bytes_per_image(arch, 14, 448, True, act_bytes=4)
    b16, post16 = bench.algorithmic_bytes_per_image(arch, 14, 448, True, act_bytes=2)
    assert post16 == post32                              # the AE stage reads fp32 head outputs in both modes
    fixed = 4 * (3 * 448 * 448 + 14 * 112 * 112 + 14 * 224 * 224 + 14 * 112 * 112)   # image + fp32 head outputs
    assert f32 - fixed == 2 * (b16 - fixed)


def test_pmc_traffic_lookup_is_keyed_by_configuration(bench):
    """A traffic figure is only quoted for the (arch, size, batch, storage) it was measured on; any other
    configuration gets (None, None) instead of another workload's number (measurement hygiene)."""
    xs = {'arch': 'search-XS', 'size': 256, 'batch': 64, 'storage': 'f32'}
    sb = {'arch': 'search-S', 'size': 448, 'batch': 32, 'storage': 'bf16'}
    per_launch, src = bench.pmc_traffic('stem_kernel', 1, xs)                 # (round 3's unfused stem: an older file)
    assert per_launch and per_launch > 1e6 and src.startswith('profiles/r0') and '_traffic' in src
    # the newest file that holds the kernel wins: the one-launch stem and the three mb16 launches per forward (round 5's
    # passes, measured with the AE stage on the mid path: no tta_project2x / peaks_topk_vec in that file)
    per_launch, src = bench.pmc_traffic('stem4_kernel', 1, xs)
    assert 2.0e8 < per_launch < 2.6e8 and src.startswith('profiles/r06_traffic.json@')
    per_launch, src = bench.pmc_traffic('mb16_kernel', 3, xs)
    assert 7e7 < per_launch < 1e8 and src.startswith('profiles/r06_traffic.json@')
    # a kernel that left the path is still quoted from the newest file that measured it
    per_launch, src = bench.pmc_traffic('tta_project2x_kernel', 1, xs)
    assert per_launch and src.startswith('profiles/r04_traffic_final.json@')
    # the PMC summary keeps the template arguments of the dw* kernels; the family name still resolves
    per_launch, src = bench.pmc_traffic('dwpw_kernel', 1, xs)
    assert per_launch and per_launch > 1e8
    assert bench.pmc_traffic('no_such_kernel', 1, xs) == (None, None)
    t, src = bench.pmc_traffic('dwb_kernel<7,1>', 31, sb)          # left the path in round 3: an older file still holds it
    assert t and '_bf16' in src
    # round 6: the fused bf16 stem is in the newest S@448 pass, a quarter of the unfused chain's 1.03 GB
    t, src = bench.pmc_traffic('stem4_kernel', 1, sb)
    assert 2.0e8 < t < 3.0e8 and src.startswith('profiles/r06_traffic_bf16_S448.json@')
    # never across configurations: S@448 fp32 has no committed PMC pass
    assert bench.pmc_traffic('pw3_kernel', 1, dict(sb, storage='f32')) == (None, None)
    # M@512 bf16 (BASELINE config 5 per GPU) has its own passes since round 5; its fused path has no unfused 7x7 depthwise
    mb = dict(sb, arch='search-M', size=512)
    assert bench.pmc_traffic('dwb_kernel<7,1>', 31, mb) == (None, None)
    t, src = bench.pmc_traffic('mbtb_kernel', 31, mb)
    assert t and src.startswith('profiles/r06_traffic_bf16_M512.json@')       # newest pass of that shape (round 6)
    assert bench.pmc_traffic('dwb_kernel<7,1>', 31, xs) == (None, None)


def test_flop_price_uses_one_peak_per_flop_class(bench):
    """VERDICT r03, measurement hygiene 9: a bf16 line once reported frac 1.033 because bf16-MFMA FLOPs and fp32-VALU
    FLOPs were priced together against the 157.3 TF fp32 peak.  price_flops() prices the depthwise FMAs at the vector
    peak and the 1x1 / deconv products at the matrix-core peak of the storage mode; the time floor is their SUM (the
    two pipes take turns in a fused block), so the fraction cannot exceed 1 for any physically possible launch."""
    # fp32 storage: both classes at 157.3 TF -> the old single-peak number
    p = bench.price_flops(4.586e9 * 19, 1.5e9 * 19, 1.28, 'f32')
    assert abs(p['frac_flops'] - 4.586e9 * 19 / 1.28e-3 / 157.3e12) < 1e-3
    assert p['floor_ms_max'] <= p['floor_ms_sum']
    # the offending line (profiles/r03_bench_n1_M512_b32_bf16.json): mbtb launches at ~162 TF of mixed FLOPs
    line = json.loads(open(os.path.join(ROOT, 'profiles', 'r03_bench_n1_M512_b32_bf16.json')).read().strip().splitlines()[-1])
    rl = line['roofline']
    assert rl['frac'] > 1.0                                    # what round 3 printed
    fl = rl['alg_flops_per_launch'] * rl['launches']
    ms = rl['avg_launch_us'] * rl['launches'] * 1e-3
    # M@512: 7x7 depthwise taps are ~19 % of a block's FLOPs (6C*49 of 6C*49 + 2*C*6C per pixel at C = 48..120)
    for share in (0.1, 0.19, 0.3, 0.6):
        q = bench.price_flops(fl, share * fl, ms, 'bf16')
        assert 0.0 < q['frac_flops'] <= 1.0, (share, q)
        assert q['mfma_peak_tflops'] == bench.BF16_MFMA_PEAK_TFLOPS and q['valu_peak_tflops'] == bench.FP32_PEAK_TFLOPS
    # a launch of pure vector FLOPs can reach at most the vector peak
    assert bench.price_flops(157.3e12 * 1e-3, 157.3e12 * 1e-3, 1.0, 'bf16')['frac_flops'] == 1.0


def test_baseline_config_presets(bench):
    """--config N = BASELINE.json's configs[N-1] (config 5 per GPU: 256 images over 8 GPUs)."""
    c = bench.CONFIGS
    assert c[3] == dict(arch='search-XS', size=256, batch=64, storage='f32') and c[2] == c[3]
    assert c[4] == dict(arch='search-S', size=448, batch=32, storage='bf16')
    assert c[5] == dict(arch='search-M', size=512, batch=32, storage='bf16')
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))['configs']
    assert 'search-S' in base[3] and '448' in base[3] and 'batch=32' in base[3] and 'bf16' in base[3]
    assert 'search-M' in base[4] and '512' in base[4] and 'batch=256' in base[4] and '8' in base[4]


def test_oks_metric_basics():
    """oracle/oks.py (the similarity the bf16 path is reported in): identical records score 1, a missing person 0, a
    3-pixel shift of a ~65 x 90 px person ~0.97 with CrowdPose's sigmas; greedy matching is one-to-one."""
    import numpy as np
    from oracle import oks
    r = np.zeros((14, 5), np.float32)
    r[:, 0] = np.arange(14) * 5
    r[:, 1] = np.arange(14) * 7
    r[:, 2] = 0.5
    assert oks.person_oks(r, r) == 1.0
    c = r.copy()
    c[:, 0] += 3
    assert 0.95 < oks.person_oks(r, c) < 0.99
    miss = r.copy()
    miss[7:, 2] = 0                                             # the candidate lost half of the joints
    assert abs(oks.person_oks(r, miss) - 0.5) < 1e-6
    far = r.copy()
    far[:, :2] += 500
    assert oks.image_oks([r, far], [far]) == [0.0, 1.0]         # one candidate serves one reference person
    assert oks.image_oks([r], []) == [0.0] and oks.image_oks([], [r]) == []
    s = oks.summary([1.0, 1.0, 0.5, 0.0])
    assert s['persons'] == 4 and s['mean'] == 0.625 and s['min'] == 0.0


def test_default_line_names_its_baseline_config(bench):
    """VERDICT r04 hygiene: the default run IS BASELINE config 3 (full path on the config-2/3 workload) and must say so;
    an explicit --config wins; a workload that is no BASELINE config says None."""
    assert bench.baseline_config_of('search-XS', 256, 64, 'f32') == 3
    assert bench.baseline_config_of('search-XS', 256, 64, 'f32', asked=2) == 2
    assert bench.baseline_config_of('search-S', 448, 32, 'bf16') == 4
    assert bench.baseline_config_of('search-M', 512, 32, 'bf16') == 5
    assert bench.baseline_config_of('search-S', 448, 32, 'f32') is None


def test_path_note_and_roofline_quote_the_same_traffic_file(bench):
    """One traffic file per line: path_roofline.note / traffic_source come from traffic_file(), roofline.traffic_source from
    pmc_traffic(); for a configuration with a committed PMC pass both must name the same file, and the step total is the sum
    over all kernels of that file (round 4: 4.18 GB for the headline = 0.17 of the HBM peak at 3.08 ms)."""
    xs = {'arch': 'search-XS', 'size': 256, 'batch': 64, 'storage': 'f32'}
    src, t = bench.traffic_file(xs)
    assert src and src.startswith('profiles/r0') and '@' in src
    total = sum(v['hbm_bytes_per_forward'] for v in t['kernels'].values())
    assert 3.0e9 < total < 5.5e9
    # round 5: merge + AE stage <= 1.2 GB per batch (round 4: 1.88 GB)
    ae = sum(v['hbm_bytes_per_forward'] for k, v in t['kernels'].items()
             if k.split('_')[0] in ('tta', 'peaks', 'refine', 'adjust', 'group', 'final', 'zero'))
    assert src.startswith('profiles/r06_traffic.json@') and ae < 1.2e9, ae
    dom = max((k for k in t['kernels']), key=lambda k: t['kernels'][k]['hbm_bytes_per_forward'])
    assert bench.pmc_traffic(dom, 1, xs)[1] == src
    assert bench.traffic_file(dict(xs, arch='search-L')) == (None, None)


def test_affinity_is_off_for_a_single_rank_and_never_raises(bench):
    """gpu_affinity(): world 1 must not pin (the CPU baseline needs the host's cores); with world > 1 and no GPU / no sysfs
    entry it reports why it did not pin instead of failing the bench."""
    import os
    before = os.sched_getaffinity(0)
    assert bench.gpu_affinity(0, 1) == {'pinned': False, 'why': 'single rank'}
    r = bench.gpu_affinity(1, 2)                   # no GPU in this container: best effort, stated
    assert r['pinned'] is False and r['why']
    assert os.sched_getaffinity(0) == before


def test_committed_default_line_carries_the_contract_and_configs_4_and_5():
    """The line `python bench.py` printed on the round's final build (profiles/r05_last_bench_n1.json): the driver's
    contract keys, BASELINE config 3 named, roofline + cpu_baseline present, one traffic file per line, no field labelled
    plain `gbps`, and BASELINE configs 4 / 5 attached with their own step time, roofline, parity and OKS (VERDICT r04 items
    1c and 7)."""
    line = json.loads(open(os.path.join(ROOT, 'profiles', 'r05_last_bench_n1.json')).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['config']['baseline_config'] == 3 and line['dtype'] == 'f32' and line['vs_baseline'] is None
    assert abs(line['value'] - 64 * line['steps'] / (line['ms_per_step'] * 1e-3 * line['steps'])) < 1.0
    assert line['graph_replay'] is True and line['parity_checked'] is True and line['parity']['images'] == 64
    rl, pr = line['roofline'], line['path_roofline']
    assert rl['traffic_source'] == pr['traffic_source'] and rl['traffic_source'].split('@')[0] in pr['note']
    assert abs(rl['frac'] - max(rl['frac_alg_bytes'], rl['frac_flops'])) < 1e-9 and 0 < rl['frac'] <= 1
    assert 'gbps' not in rl and all('gbps' not in v for v in line['kernels'].values())
    assert all(v['hbm_gbps'] is None or v['hbm_gbps'] < 8000.0 for v in line['kernels'].values())
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] >= 1
    assert line['per_rank']['affinity'] == [{'pinned': False, 'why': 'single rank'}]
    for n, (arch, dtype) in {'4': ('S@448', 'bf16'), '5': ('M@512', 'bf16')}.items():
        c = line['configs'][n]
        assert 'error' not in c and arch in c['workload'] and c['dtype'] == dtype and c['steps'] == line['steps']
        assert c['graph_replay'] is True and c['ms_per_step'] > 0 and 0 < c['path_frac'] < 1 and 0 < c['frac_flops'] < 1
        assert c['roofline']['kernel'] == 'mbtb_kernel' and 0 < c['roofline']['frac'] <= 1
        assert c['parity']['ok'] is True and c['parity']['records_identical_to_oracle_parser'] is True
        assert c['parity']['oks']['mean'] >= 0.999 and c['parity']['heatmap_err'] <= c['parity']['tolerance']


def test_committed_round6_line_carries_the_honesty_fields():
    """The line `python bench.py` printed on round 6's last kernel build (profiles/r06_bench_n1.json): on top of the contract
    keys checked on round 5's line above -- the second, 200-step timed region of the same run, batch-1 / batch-8 latency, the
    CU occupancy of the dominant family from the launch geometry, the second FLOP yardstick, and child lines that checked 16
    of their 32 images."""
    line = json.loads(open(os.path.join(ROOT, 'profiles', 'r06_bench_n1.json')).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['config']['baseline_config'] == 3 and line['dtype'] == 'f32' and line['vs_baseline'] is None
    assert abs(line['value'] - 64e3 / line['ms_per_step']) < 1.0
    assert 0 < line['ms_per_step_200'] < 1.05 * line['ms_per_step']              # a longer region of the same run, not a better box
    assert 0 < line['latency_ms_batch1'] < line['latency_ms_batch8'] < line['latency_ms_single_batch_graph'] < line['latency_ms_single_batch']
    rl, pr = line['roofline'], line['path_roofline']
    assert rl['kernel'] == 'mb16_kernel' and rl['cus_total'] == 256 and 0 < rl['cus_occupied'] <= 256
    assert abs(rl['frac_flops_per_occupied_cu'] - rl['frac_flops'] * 256 / rl['cus_occupied']) < 2e-3
    assert 0 < pr['frac_flops_bf16x3'] < pr['frac_flops'] < 1                  # the same floor priced at a higher matrix peak
    assert rl['traffic_source'].startswith('profiles/r06_traffic.json@') and rl['traffic'] > 0
    assert line['parity']['images'] == 64 and line['parity']['ok'] is True
    for n in ('4', '5'):
        c = line['configs'][n]
        assert 'error' not in c and c['dtype'] == 'bf16' and c['graph_replay'] is True
        assert c['parity']['ok'] is True and c['parity']['images'] == 16 and c['parity']['records_identical_to_oracle_parser'] is True
        assert 0 < c['ms_per_step_200'] < 1.05 * c['ms_per_step'] and c['wall_s'] < 240


def test_child_line_condenser_on_a_committed_config4_line(bench):
    """bench.condense_child_line (what the default run attaches under `configs`) applied to a full `--config 4` line of the
    same round: every figure is carried over unchanged, nothing is recomputed."""
    full = json.loads(open(os.path.join(ROOT, 'profiles', 'r05_final_bench_n1_S448_b32_bf16.json')).read().strip().splitlines()[-1])
    c = bench.condense_child_line(full, 4, full['steps'], full['warmup'], 12.3)
    assert c['ms_per_step'] == full['ms_per_step'] and c['value'] == full['value'] and c['dtype'] == 'bf16'
    assert c['path_frac'] == full['path_roofline']['frac'] and c['frac_flops'] == full['path_roofline']['frac_flops']
    assert c['roofline']['kernel'] == full['roofline']['kernel'] == 'mbtb_kernel'
    assert c['roofline']['frac'] == full['roofline']['frac'] and c['roofline']['traffic'] == full['roofline']['traffic']
    assert c['kernels_ms'] == {k: v['ms_per_step'] for k, v in full['kernels'].items()}
    assert c['parity']['oks'] == full['parity']['p3_vs_pure_cpu_pipeline']['oks_vs_cpu_persons']
    assert c['parity']['heatmap_err'] == full['parity']['heatmap_tag_max_abs_err'] and c['wall_s'] == 12.3
    assert '--config 4' in c['command'] and 'S@448' in c['workload']


def test_cus_occupied_and_the_second_flop_yardstick(bench):
    """Round 6: `roofline.cus_occupied` = the CUs a launch can hold at all -- min(256, grid /
    workgroups per CU by the occupancy query) -- so that 0.43-of-the-chip is readable as 0.86-of-half-of-it; and the bf16x3
    yardstick prices the MFMA class of the fp32 path at what the bf16 pipe delivers of fp32-exact products (2 500 / 6 TF)."""
    assert bench.cus_occupied(128, 1) == 128.0                 # mb16_kernel at batch 64: one workgroup per image + mirror
    assert bench.cus_occupied(3136, 1) == 256.0 and bench.cus_occupied(512, 2) == 256.0
    assert bench.cus_occupied(100, 3) == 34.0 and bench.cus_occupied(0, 0) == 256.0 and bench.cus_occupied(5, 0) == 5.0
    assert abs(bench.BF16X3_EQUIV_TFLOPS - 2500.0 / 6.0) < 1e-9
    fl, fv, ms = 226.0e9, 81.9e9, 2.949                        # the round-5 headline: 81.9 GF VALU + 144.1 GF MFMA class
    a = bench.price_flops(fl, fv, ms, 'f32')
    b = bench.price_flops(fl, fv, ms, 'f32', mfma_peak=bench.BF16X3_EQUIV_TFLOPS)
    assert abs(a['frac_flops'] - 0.487) < 2e-3 and abs(b['frac_flops'] - 0.294) < 2e-3     # VERDICT r05's two numbers
    assert b['flops_valu'] == a['flops_valu'] and b['mfma_peak_tflops'] < a['mfma_peak_tflops'] * 3
