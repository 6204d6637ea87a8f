#!/usr/bin/env python3
"""The two K applies inside S three ways: one after the other, side by side on two HIP
streams (what SchurMPI does, MultiGrid.apply_pair), and BATCHED -- the two right-hand
sides as column ranges of one slab of 2 ld columns, one V-cycle chain with half as many
launches of twice the size (time slices are independent in every space operator, so the
results are the same column by column)."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time)
K = h.Kinv_x
n_loc = h.N
ld = n_loc + (n_loc & 1)
both = torch.rand((h.M, 2 * ld), dtype=torch.float64, device='cuda')
both[:, n_loc:ld] = 0
both[:, ld + n_loc:] = 0
u1, u2 = both[:, :ld].contiguous(), both[:, ld:].contiguous()
out_both = torch.empty_like(both)


def sequential():
    return K.apply(u1, n_loc=n_loc), K.apply(u2, n_loc=n_loc)


def two_streams():
    return K.apply_pair(u1, u2, n_loc=n_loc)


def batched():
    return K.apply(both, out=out_both, n_loc=2 * ld)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.iters)
    return best


a, b, c = timed(sequential), timed(two_streams), timed(batched)
y1, y2 = two_streams()
yb = batched()
torch.cuda.synchronize()
same = bool(torch.equal(yb[:, :n_loc], y1[:, :n_loc]) and torch.equal(yb[:, ld:ld + n_loc], y2[:, :n_loc]))
print('J_time=%d J_space=%d (%d steps): two K applies one after the other %.3f ms | on two streams %.3f ms | '
      'batched in one slab of %d columns %.3f ms  (bit-identical: %s)' % (args.J_time, args.J_space, n_loc, a, b, 2 * ld, c, same))
