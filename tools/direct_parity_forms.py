#!/usr/bin/env python3
"""History of the direct-preconditioned solve of config 1 against the oracle's, for the three
forms of InvLinOp's apply: dense top block (default), every level by itself, SuperLU on the host."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source.linalg import PCG  # noqa: E402
from source.linop import InvLinOp  # noqa: E402

g = np.load(os.path.join(REPO, 'tests', 'golden', 'o1_pcg_square_J3_J6_direct.npz'))
ref = np.asarray(g['hist'])
default_block = InvLinOp.TOP_BLOCK
for name, dense_top, host, block in (('dense top, whole block inverted', True, False, default_block),
                                     ('dense top, blocks of 256', True, False, 256),
                                     ('every level by itself', False, False, default_block),
                                     ('SuperLU on the host', True, True, default_block)):
    InvLinOp.dense_top, InvLinOp.host_solve, InvLinOp.TOP_BLOCK = dense_top, host, block
    try:
        h = hm.HeatEquationMPI(J_space=6, J_time=3, precond='direct')
    finally:
        InvLinOp.dense_top, InvLinOp.host_solve, InvLinOp.TOP_BLOCK = True, False, default_block
    for op in [h.Kinv_x] + list(h.C_j):
        op.host_solve = host
    hist = []
    w, it = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
    rel = np.abs(np.asarray(hist) / ref - 1.0)
    print('%-24s iterations %d  per entry %s  max %.1e' % (name, it, ' '.join('%.0e' % v for v in rel), rel.max()))
