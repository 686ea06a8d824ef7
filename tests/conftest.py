import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def g_csmri():
    return golden('csmri_setup.npz')


@pytest.fixture(scope='session')
def g_denoise():
    return golden('denoise.npz')


@pytest.fixture(scope='session')
def g_traces64():
    return golden('traces64.npz')


@pytest.fixture(scope='session')
def g_traces256():
    return golden('traces256.npz')


@pytest.fixture(scope='session')
def g_traces256_full():
    return golden('traces256_full.npz')


@pytest.fixture(scope='session')
def g_deblur():
    return golden('deblur.npz')


@pytest.fixture(scope='session')
def g_pr():
    return golden('pr.npz')


@pytest.fixture(scope='session')
def g_psnr():
    return golden('psnr.npz')


@pytest.fixture(scope='session')
def g_r2():
    return golden('r2_fixtures.npz')
