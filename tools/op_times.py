#!/usr/bin/env python3
"""Per-operator device times at the bench configuration (W, WT, S, P, kron,
BLAS-1), the analogue of the reference's heateq_mpi_timing.py."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from bench import seeded_slab  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--problem', default='square')
ap.add_argument('--alternate', type=int, default=1)
ap.add_argument('--fuse', type=int, default=1)
ap.add_argument('--coarse-rows', type=int, default=4096)
ap.add_argument('--coarse-pairs', type=int, default=0)
ap.add_argument('--fuse-restrict', type=int, default=1)
ap.add_argument('--zero-start', type=int, default=1)
ap.add_argument('--arithmetic', default='accurate', help="HeatEquationMPI's arithmetic mode (accurate = the default, fast, reference)")
ap.add_argument('--tune', default='', help='extra stk_set_tuning keys: key=value,...')
ap.add_argument('--only', default='', help='comma-separated subset of the rows (W,WT,S,P,Kinv,A_x,A_x_packed,axpy,dot)')
ap.add_argument('--wavelettransform', default=None, help="HeatEquationMPI's wavelet mode (composite, original, interleaved)")
args = ap.parse_args()
from source import _lib  # noqa: E402
_lib.check(_lib.lib().stk_set_tuning(b'rows_alternate', args.alternate))
_lib.check(_lib.lib().stk_set_tuning(b'mg_fuse_coarse', args.fuse))
_lib.check(_lib.lib().stk_set_tuning(b'mg_coarse_max_rows', args.coarse_rows))
_lib.check(_lib.lib().stk_set_tuning(b'mg_coarse_pairs', args.coarse_pairs))
_lib.check(_lib.lib().stk_set_tuning(b'mg_fuse_restrict', args.fuse_restrict))
_lib.check(_lib.lib().stk_set_tuning(b'mg_zero_start', args.zero_start))
for kv in filter(None, args.tune.split(',')):
    k, v = kv.split('=')
    _lib.check(_lib.lib().stk_set_tuning(k.encode(), int(v)))
kw = {} if args.wavelettransform is None else {'wavelettransform': args.wavelettransform}
h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time, problem=args.problem, arithmetic=args.arithmetic, **kw)
print('arithmetic=%s' % args.arithmetic)
dd = h.dofs_distr
x = KronVectorMPI(dd, seeded_slab(dd.t_begin, dd.t_end, h.M))
y = x.copy()
nb = 8.0 * h.N * h.M


def timeit(fn, n=args.iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = [('W', lambda: h.W @ x, 2), ('WT', lambda: h.WT @ x, 2), ('S', lambda: h.S @ x, None),
        ('P', lambda: h.P @ x, None), ('Kinv', lambda: h.Kinv_x.apply(x.buf, n_loc=x.n_loc), None),
        ('A_x', lambda: h.CAC_j[0].linops[1].apply(x.buf, n_loc=x.n_loc), 2),
        ('axpy', lambda: y.__iadd__(0.5 * x), 3), ('dot', lambda: x.dot(y), 2)]
if getattr(h.S, 'ell', None) is not None and h.S.ell.packed_for(x.n_loc).ok:
    pk, out_ = h.S.ell.packed_for(x.n_loc), torch.empty_like(x.buf)
    rows.insert(6, ('A_x_packed', lambda: pk.apply([(None, 1)], x.buf, None, x.n_loc, x.ld, 0.0, out_), 2))
only = set(filter(None, args.only.split(',')))
for name, fn, passes in rows:
    if only and name not in only:
        continue
    ms = timeit(fn)
    extra = '' if passes is None else '  %.0f GB/s (%d vector passes)' % (passes * nb / ms / 1e6, passes)
    print('%-10s %9.3f ms%s' % (name, ms, extra))
