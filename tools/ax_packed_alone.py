#!/usr/bin/env python3
"""I kron A_x three ways at one slab shape: the row engine on A_x's own 5-slot copy (what
BlockDiagMPI runs), the packed (M_x, A_x) union stream as one term (round 4: slower), and
a packed plan of A_x ALONE (pairs of neighbouring rows share two of their five columns:
8 gathers for two rows)."""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib  # noqa: E402
from source.assembly import space_matrices  # noqa: E402
from source.linop import EllMatrices, SpaceMatrix  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--n_loc', type=int, default=65)
ap.add_argument('--problem', default='square')
args = ap.parse_args()
mesh = problem_helper(args.problem, J_space=args.J_space, J_time=2)[0]
M_x, A_x = space_matrices(mesh)
M = A_x.shape[0]
n_loc = args.n_loc
ld = n_loc + (n_loc & 1)
x = torch.rand((M, ld), dtype=torch.float64, device='cuda')
x[:, n_loc:] = 0
y0, y1, y2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
rows = SpaceMatrix(A_x)
union = EllMatrices([M_x, A_x], [M_x]).packed_for(n_loc)
t = time.time()
alone_plan = EllMatrices([A_x], [M_x])
alone = alone_plan.packed_for(n_loc)
torch.cuda.synchronize()
print('plan of A_x alone: %.2f s, ok=%s, K=%s, rows per unit %s, %d slot rows' % (
    time.time() - t, alone.ok, getattr(alone, 'K', None), alone.rows_per_unit, getattr(alone, 'n_units', 0)))


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


a = timed(lambda: rows.apply(x, out=y0, n_loc=n_loc))
b = timed(lambda: union.apply([(None, 1)], x, None, n_loc, ld, 0.0, y1))
c = timed(lambda: alone.apply([(None, 0)], x, None, n_loc, ld, 0.0, y2)) if alone.ok else float('nan')
nb = 16.0 * n_loc * M + 12.0 * A_x.nnz + 4.0 * (M + 1)
print('%s J_space=%d n_loc=%d: row engine %.3f ms | union stream %.3f ms (equal: %s) | A_x alone, packed %.3f ms '
      '(equal: %s)  [%.0f MB algorithmic: %.0f / %.0f / %.0f GB/s]' % (
          args.problem, args.J_space, n_loc, a, b, bool(torch.equal(y0, y1)), c,
          bool(torch.equal(y0, y2)) if alone.ok else None, nb / 1e6, nb / a / 1e6, nb / b / 1e6, nb / c / 1e6))
