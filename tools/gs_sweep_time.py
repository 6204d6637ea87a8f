#!/usr/bin/env python3
"""Time of the finest-level Gauss-Seidel sweeps alone (stk_mg_smooth), per row,
for several mesh sizes at the same slab length: how much a working set that
fits the Infinity Cache is worth."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source.assembly import space_matrices  # noqa: E402
from source.mesh import construct_2d_square_mesh  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily  # noqa: E402

n_loc = int(sys.argv[2]) if len(sys.argv) > 2 else 65
ld = n_loc + (n_loc & 1)
for J in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '7,8,9').split(',')]:
    mesh, _ = construct_2d_square_mesh(J)
    M_x, A_x = space_matrices(mesh)
    h = MeshHierarchy(mesh)
    for name, mg in (('A-only (3 groups)', MultiGrid(A_x, h, 3, 2)),
                     ('M+A (4 groups)', MultiGrid((M_x + A_x).tocsr(), h, 3, 2))):
        M = A_x.shape[0]
        u = torch.zeros((M, ld), dtype=torch.float64, device='cuda')
        f = torch.rand((M, ld), dtype=torch.float64, device='cuda')
        f[:, n_loc:] = 0
        for bw in (False, True):
            for _ in range(3):
                mg.smooth(h.J, u, f, 3, backward=bw, n_loc=n_loc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 10
            for _ in range(reps):
                mg.smooth(h.J, u, f, 3, backward=bw, n_loc=n_loc)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps / 3
            print('J_space=%d %s %s: %.3f ms per sweep, %.3f ns per row' %
                  (J, name, 'bwd' if bw else 'fwd', ms, ms * 1e6 / M))
