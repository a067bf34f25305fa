import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    d = os.path.join(ROOT, 'tests', 'golden')
    return {n: np.load(os.path.join(d, n + '.npz')) for n in
            ('fdconv_layers', 'group_maps', 'fbm_masks', 'rrdbnet_fdconv_1block', 'realesrganer')}


@pytest.fixture(scope='session')
def golden_r2():
    """Round-2 fixtures (tests/golden/make_golden_r2.py): reference gradients, converter output, SRVGG outputs."""
    import numpy as np
    d = os.path.join(ROOT, 'tests', 'golden')
    return {n: np.load(os.path.join(d, n + '.npz')) for n in
            ('fdconv_train_grads', 'rrdbnet_train_step', 'convert_ref', 'srvgg', 'fdconv_layers_r2')}
