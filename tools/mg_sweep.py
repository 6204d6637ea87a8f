#!/usr/bin/env python3
"""The multigrid parameter sweep of the reference (source/multigrid.py:200-229):
for smoothsteps x vcycles in {1..4}^2 the condition number of K^-1 A_x by Lanczos
and the device time of one multigrid apply, on the stiffness matrix of the square.

    python tools/mg_sweep.py [--J_space 9] [--n_loc 1] [--numbering both]

--numbering both also runs the vertex numbering of rounds 1-3 (hypotenuse
midpoints LAST among a level's new vertices: 3 Gauss-Seidel dependency groups
for A_x instead of 2): the sweep order is the dof order, so the numbering is
part of the smoother -- this is what it does to kappa and to the time."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import mesh as mesh_mod  # noqa: E402
from source.assembly import space_matrices  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.lanczos import Lanczos  # noqa: E402
from source.mpi_kron import IdentityKronMatMPI  # noqa: E402
from source.mpi_vector import DofDistributionMPI, KronVectorMPI  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid, gauss_seidel_schedule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--n_loc', type=int, default=1, help='time slices the operators are applied to at once')
ap.add_argument('--numbering', default='first', choices=['first', 'last', 'both'])
ap.add_argument('--max', type=int, default=4)
args = ap.parse_args()

for first in {'first': [True], 'last': [False], 'both': [True, False]}[args.numbering]:
    mesh_mod.HYPOTENUSE_FIRST = first
    mesh, _ = mesh_mod.construct_2d_square_mesh(args.J_space)
    M_x, A_x = space_matrices(mesh)
    hierarchy = MeshHierarchy(mesh)
    groups = len(gauss_seidel_schedule(A_x.indptr, A_x.indices)[0]) - 1
    print('A_x %s, hypotenuse midpoints %s: %d Gauss-Seidel dependency groups'
          % (A_x.shape, 'first' if first else 'last', groups), flush=True)
    dd = DofDistributionMPI(MPI.COMM_WORLD, args.n_loc, A_x.shape[0])
    A = IdentityKronMatMPI(dd, A_x)
    start = np.random.RandomState(3).rand(args.n_loc, A_x.shape[0]) * 2.0 - 1.0
    for smoothsteps in range(1, args.max + 1):
        for vcycles in range(1, args.max + 1):
            mg = MultiGrid(A_x, hierarchy, smoothsteps=smoothsteps, vcycles=vcycles)
            P = IdentityKronMatMPI(dd, mg)
            lz = Lanczos(A, P, w=KronVectorMPI(dd, start.copy()))
            x = KronVectorMPI(dd, start.copy())
            P @ x
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                P @ x
            e1.record()
            e1.synchronize()
            print('smoothsteps=%d vcycles=%d time_per_apply=%.3f ms lanczos: lmax=%.6f lmin=%.6f kappa=%.4f (%d its)'
                  % (smoothsteps, vcycles, e0.elapsed_time(e1) / 5, lz.lmax, lz.lmin, lz.lmax / lz.lmin,
                     lz.iterations), flush=True)
            del mg, P
