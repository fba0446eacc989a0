import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def ns():
    """the ``from sdf import *`` namespace of the implementation under test"""
    import sdf_amd
    return {k: getattr(sdf_amd, k) for k in dir(sdf_amd) if not k.startswith('_')}


@pytest.fixture(scope='session')
def golden_values():
    return np.load(os.path.join(GOLDEN, 'values.npz'))


@pytest.fixture(scope='session')
def oracle_lib():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope='session')
def eng():
    """the HIP engine; GPU tests fail (not skip) if the extension is missing or no device"""
    try:                      # torch bundles its own HIP runtime: when a test uses both, torch's
        import torch          # must be the one that initialises first
        torch.cuda.is_available()
    except Exception:
        pass
    from sdf_amd import engine
    return engine.get_engine(0)


def value_tolerance(ref, P):
    """float64 parity bound for SDF values: a few ulp of the larger of |value| and |point|
    (cancellation: a distance is a difference of coordinate-sized numbers).  The slack over
    bit-equality covers the BLAS / libm dependent operations of the reference (np.dot,
    sin/cos/arctan2/hypot/power), see DESIGN.md 'numerics'."""
    scale = np.maximum(np.abs(ref), np.abs(P).max(axis=1))
    return 16 * np.spacing(scale)


def soup_key(points):
    """order-invariant canonical form of a triangle soup: rows of 9 sorted lexicographically"""
    t = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 9)
    return t[np.lexsort(t.T[::-1])]
