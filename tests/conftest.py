"""Shared test configuration.

`gpu` marks every test that needs a real MI355X; everything else runs on CPU.
The product package lives in ``spacetime-fullgrid-parallel_amd/`` and mirrors
the reference's ``source`` package, so that directory is put on sys.path.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'spacetime-fullgrid-parallel_amd')
GOLDEN = os.path.join(REPO, 'tests', 'golden')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
if os.environ.get('STK_TEST_HEAP') == '1':
    # the suite under the allocator policy the drivers and bench.py run with
    # (source/host_malloc.py; by default a test process keeps glibc's settings and every
    # HeatEquationMPI scopes the heap mode to its own set-up)
    from source.host_malloc import keep_to_the_heap
    keep_to_the_heap()

import numpy as np  # noqa: E402
import pytest  # noqa: E402
import scipy.sparse as sp  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def csr_from(g, prefix):
    return sp.csr_matrix((g[prefix + '_data'], g[prefix + '_indices'],
                          g[prefix + '_indptr']),
                         shape=tuple(g[prefix + '_shape']))


def problem_from(g):
    """The matrices stored in a g3_* fixture, as the dict HeatEquationOracle
    and the device wiring take."""
    mats = {k: csr_from(g, k) for k in ('A_t', 'L_t', 'M_t', 'G_t', 'M_x',
                                         'A_x')}
    mats['P_mats'] = [csr_from(g, 'P%d' % j) for j in range(int(g['nP']))]
    mats['u0_t'] = g['u0_t']
    mats['u0_x'] = g['u0_x']
    return mats


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(params=['g3_square', 'g3_lshape', 'g3_square3', 'g3_cube'])
def g3(request):
    return load_golden(request.param)
