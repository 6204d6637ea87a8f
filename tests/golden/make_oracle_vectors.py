#!/usr/bin/env python3
"""Mid-size PCG trajectory from the CPU ORACLE (not the reference: the
reference's pure-Python smoother would need hours at this size).  The oracle
is pinned to the reference by tests/test_oracle_golden.py; this fixture extends
the GPU parity check to N = 33, M = 16 129.

    python tests/golden/make_oracle_vectors.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

from oracle.heat import HeatEquationOracle  # noqa: E402
from oracle.krylov import pcg  # noqa: E402
from source.assembly import (prolongation_matrices, space_load,  # noqa: E402
                             space_matrices, time_matrices)
from source.problem import problem_helper  # noqa: E402


def main():
    J_time, J_space = 5, 6
    mesh, _, tmesh, data, _ = problem_helper('square', J_space, J_time)
    A_t, L_t, M_t, G_t, u0_t = time_matrices(tmesh)
    M_x, A_x = space_matrices(mesh)
    mats = dict(A_t=A_t, L_t=L_t, M_t=M_t, G_t=G_t, M_x=M_x, A_x=A_x,
                P_mats=prolongation_matrices(mesh), u0_t=u0_t,
                u0_x=space_load(mesh, data['u0']))
    o = HeatEquationOracle(mats, J_time)
    t0 = time.time()
    w, iters, hist = pcg(o.WT_S_W, o.P, o.rhs())
    print('oracle PCG: %d iterations in %.1f s' % (iters, time.time() - t0))
    X = np.random.RandomState(128).rand(o.N, o.M)
    np.savez_compressed(os.path.join(HERE, 'o1_pcg_J5_J6.npz'), J_time=J_time,
                        J_space=J_space, iters=iters, hist=np.array(hist),
                        w_norm=np.linalg.norm(w),
                        w_sample=w[::4, ::97].copy(),
                        SX_sample=o.S(X)[::4, ::97].copy(),
                        PX_sample=o.P(X)[::4, ::97].copy())


if __name__ == '__main__':
    main()
