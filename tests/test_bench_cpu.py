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
    per_launch, src = bench.pmc_traffic('stem_kernel', 1, xs)
    assert per_launch and per_launch > 1e6 and src.startswith('profiles/r0') and '_traffic' in src
    # the PMC summary keeps the template arguments of the dw* kernels; the family name still resolves
    per_launch, src = bench.pmc_traffic('dwpw_kernel', 1, xs)
    assert per_launch and per_launch > 1e8
    assert bench.pmc_traffic('no_such_kernel', 1, xs) == (None, None)
    t, src = bench.pmc_traffic('dwb_kernel<7,1>', 31, sb)
    assert t and '_bf16' in src
    # never across configurations: S@448 fp32 / M@512 bf16 have no committed PMC pass
    assert bench.pmc_traffic('pw3_kernel', 1, dict(sb, storage='f32')) == (None, None)
    assert bench.pmc_traffic('dwb_kernel<7,1>', 31, dict(sb, arch='search-M', size=512)) == (None, None)
    assert bench.pmc_traffic('dwb_kernel<7,1>', 31, xs) == (None, None)
