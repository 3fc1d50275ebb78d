import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MOBILE_CONFIGS = os.path.join(ROOT, 'litepose_amd', 'mobile_configs')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden.npz'))


def load_arch(name):
    import json
    with open(os.path.join(MOBILE_CONFIGS, name + '.json')) as f:
        return json.load(f)
