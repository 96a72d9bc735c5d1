import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import json
    import numpy as np
    g = os.path.join(ROOT, 'tests', 'golden')
    meta = json.load(open(os.path.join(g, 'cases.json')))
    arrays = np.load(os.path.join(g, 'cases.npz'))
    return meta, arrays


@pytest.fixture(scope='session')
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.lib()
    return oracle_lib
