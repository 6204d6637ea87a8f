#!/usr/bin/env python3
"""A handful of applies of the metric's Kronecker operator at one slab shape,
plain ELL form or packed form -- the small target the PMC passes of
tools/pmc_passes.sh profile (bench.py spends most of a pass in set-up)."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib  # noqa: E402
from source.assembly import space_matrices  # noqa: E402
from source.linop import EllMatrices  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--problem', default='square')
ap.add_argument('--n_loc', type=int, default=65)
ap.add_argument('--ghosts', type=int, default=0)
ap.add_argument('--kernels', default='plain,packed')
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--tune', default='')
args = ap.parse_args()
for kv in filter(None, args.tune.split(',')):
    k, v = kv.split('=')
    _lib.check(_lib.lib().stk_set_tuning(k.encode(), int(v)))
mesh = problem_helper(args.problem, J_space=args.J_space, J_time=2)[0]
M_x, A_x = space_matrices(mesh)
M = M_x.shape[0]
ell = EllMatrices([M_x, A_x], [M_x])
n_loc = args.n_loc
ld = n_loc + (n_loc & 1)
rng = np.random.RandomState(0)
x = torch.rand((M, ld), dtype=torch.float64, device='cuda')
x[:, n_loc:] = 0
y = torch.empty_like(x)
tri = [_lib.to_dev(rng.rand(3, n_loc)) for _ in range(2)]
g = torch.rand((2, M), dtype=torch.float64, device='cuda') if args.ghosts else None
gh = None
if g is not None:
    gh = torch.empty((M, 2), dtype=torch.float64, device='cuda')
    _lib.check(_lib.lib().stk_interleave_ghosts(_lib.stream(), M, _lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(gh)))
lo, hi = (g[0], g[1]) if g is not None else (None, None)
for name in args.kernels.split(','):
    for _ in range(args.reps):
        if name == 'plain':
            ell.apply([(tri[0], 0, x, lo, hi), (tri[1], 1, x, lo, hi)], n_loc, ld, 0.0, y)
        else:
            ell.packed.apply([(tri[0], 0), (tri[1], 1)], x, gh, n_loc, ld, 0.0, y)
    torch.cuda.synchronize()
print('done', M, n_loc)
if 'packed' in args.kernels.split(','):
    # what the PMC record of this run is stamped with (tools/pmc_traffic.py reads the line
    # from the pass's log): the hash of the kernel's sources and of the plan it streamed
    import bench  # noqa: E402
    print('fingerprint source_sha=%s plan_sha=%s' % (bench.kernel_source_sha(), bench.plan_sha(ell.packed)))
