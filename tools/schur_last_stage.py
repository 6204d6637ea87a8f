#!/usr/bin/env python3
"""The last stage of SchurMPI, (I kron M_x) v1 + (I kron A_x) v2 + (G_t kron M_x) x
(reference heateq_mpi.py:166-181), timed alone: the plain sliced-ELL form
(stk_kron_ell_apply, three inputs) against the packed slot stream with an input per
term (stk_kron_pack_apply_multi), results compared bit for bit."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--tune', default='')
args = ap.parse_args()
for kv in filter(None, args.tune.split(',')):
    k, v = kv.split('=')
    _lib.check(_lib.lib().stk_set_tuning(k.encode(), int(v)))
h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time)
S = h.S
n_loc, ld = h.rhs.n_loc, h.rhs.ld
x = h.rhs.buf
v1, v2 = torch.rand_like(x), torch.rand_like(x)
v1[:, n_loc:] = 0
v2[:, n_loc:] = 0
y_plain, y_pack = torch.empty_like(x), torch.empty_like(x)
packed = S.ell.packed_for(n_loc)


def plain():
    S.ell.apply([(None, 0, v1, None, None), (None, 1, v2, None, None), (S.tG, 0, x, None, None)], n_loc, ld, 0.0, y_plain)


def multi():
    packed.apply_multi([(None, 0, v1), (None, 1, v2), (S.tG, 0, x)], n_loc, ld, 0.0, y_pack)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def two_terms():  # without the G_t term: what its turn costs
    packed.apply_multi([(None, 0, v1), (None, 1, v2)], n_loc, ld, 0.0, y_pack)


nbytes = 8.0 * h.M * (3 * n_loc + 1) + 12.0 * (2 * h.M_x.nnz + h.A_x.nnz) + 4.0 * 3 * (h.M + 1)
for rep in range(2):
    a, b = timed(plain), timed(multi)
print('two-input terms alone: %.3f ms' % timed(two_terms))
multi()
print('J_time=%d J_space=%d: plain sliced-ELL form %.3f ms, packed with an input per term %.3f ms '
      '(%.0f GB/s of %.0f MB algorithmic, %.1f %% of 8 TB/s); bit-identical: %s'
      % (args.J_time, args.J_space, a, b, nbytes / b / 1e6, nbytes / 1e6, nbytes / b / 1e6 / 80.0,
         bool(torch.equal(y_plain, y_pack))))
