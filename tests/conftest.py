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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle front-end (test infrastructure)."""
    from oracle import kbe_oracle
    kbe_oracle.lib()
    return kbe_oracle


def bits(a):
    """fp32 array -> its bit patterns, so that comparisons are exact (and NaN-safe)."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)


def assert_bits_equal(a, b, what=''):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = int((bits(a) != bits(b)).sum())
    assert bad == 0, '%s: %d of %d elements differ bitwise (max abs %g)' % (
        what, bad, a.size, float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))))
