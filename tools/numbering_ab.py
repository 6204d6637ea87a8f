#!/usr/bin/env python3
"""Time to solution of PCG(W^T S W, P, rhs) under the two vertex numberings the
mesh generator offers (source/mesh.py HYPOTENUSE_FIRST): the hypotenuse-midpoint
class first among a level's new vertices (A_x sweeps in 2 dependency groups) or
last (3 groups).  The Gauss-Seidel sweep runs in dof order (reference
multigrid.py:89-97), so the numbering is part of the smoother: fewer groups make
an apply cheaper, a weaker smoother makes the solve longer.

    python tools/numbering_ab.py [--J_time 6 --J_space 9] [--problem square] [--orders]

S and P are timed alone as well; --orders runs all six orders of the three edge classes
(mesh.CLASS_ORDER) instead of the two settings of the flag."""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source import mesh as mesh_mod  # noqa: E402
from source.linalg import PCG  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--problem', default='square')
ap.add_argument('--orders', action='store_true',
                help='all six orders of the three edge classes instead of the two settings of the flag')
args = ap.parse_args()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, out


import itertools  # noqa: E402

cases = ([(False, p) for p in itertools.permutations(range(3))] if args.orders
         else [(True, None), (False, None)])
for first, perm in cases:
    mesh_mod.HYPOTENUSE_FIRST = first
    mesh_mod.CLASS_ORDER = perm
    if perm is not None:
        print('class order %s:' % (perm,), end=' ')
    h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time, problem=args.problem)
    x = h.rhs.copy()
    tS, _ = timed(lambda: h.S @ x, 5)
    tP, _ = timed(lambda: h.P @ x, 5)
    hist = []
    tsol, (w, iters) = timed(lambda: PCG(h.WT_S_W, h.P, h.rhs, history=hist), 3)
    n = len(hist) // 4  # history of one of the four solves
    print('hypotenuse midpoints %-5s: S %.3f ms, P %.3f ms, PCG %d iterations in %.1f ms (%.2f ms per iteration), '
          'r.Pr %.3e -> %.3e' % ('first' if first else 'last', tS * 1e3, tP * 1e3, iters, tsol * 1e3,
                                 tsol * 1e3 / iters, hist[0], hist[n - 1]), flush=True)
    del h, x, w
    torch.cuda.empty_cache()
