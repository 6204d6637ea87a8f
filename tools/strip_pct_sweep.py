#!/usr/bin/env python3
"""S and P apply times at config 3 for strip sizes per plan (stk_mg_set_option
"strip_pct": K's plans, whose applies run two at a time inside S, and the
preconditioner family's plan, whose applies run alone)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from bench import seeded_slab  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402

h = hm.HeatEquationMPI(J_space=9, J_time=6)
x = KronVectorMPI(h.dofs_distr, seeded_slab(0, h.N, h.M))


def timed(op, n=10):
    op @ x
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        op @ x
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for pct in (30, 45, 60, 80, 100):
    h.Kinv_x._dev.set_option('strip_pct', pct)
    print('K plans at %3d %% of 250 MB: S %.3f ms' % (pct, timed(h.S)), flush=True)
for pct in (100, 130, 160, 200, 240, 320):
    h.C_family._dev.set_option('strip_pct', pct)
    print('family plan at %3d %%: P %.3f ms' % (pct, timed(h.P)), flush=True)
