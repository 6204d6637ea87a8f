#!/usr/bin/env python3
"""Do the two independent multigrid applies inside S (K u1 and K u2) gain from
running side by side on two HIP streams?  Sequential on one stream against
concurrent on two (a second plan instance owns the second set of workspaces)."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source.multigrid import MultiGrid  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time)
K1 = h.Kinv_x
K2 = MultiGrid(h.A_x, h.hierarchy, smoothsteps=3, vcycles=2)
n_loc = h.N
ld = n_loc + (n_loc & 1)
u1 = torch.rand((h.M, ld), dtype=torch.float64, device='cuda')
u2 = torch.rand((h.M, ld), dtype=torch.float64, device='cuda')
u1[:, n_loc:] = 0
u2[:, n_loc:] = 0
o1, o2 = torch.empty_like(u1), torch.empty_like(u2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def sequential():
    K1.apply(u1, out=o1, n_loc=n_loc)
    K1.apply(u2, out=o2, n_loc=n_loc)


def concurrent():
    done = torch.cuda.Event()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        K1.apply(u1, out=o1, n_loc=n_loc)
    with torch.cuda.stream(s2):
        K2.apply(u2, out=o2, n_loc=n_loc)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


sequential()
a1, a2 = o1.clone(), o2.clone()
concurrent()
torch.cuda.synchronize()
print('identical results:', torch.equal(a1, o1) and torch.equal(a2, o2))
for rnd in range(3):
    print('two K applies: one stream %.3f ms, two streams %.3f ms' % (timed(sequential), timed(concurrent)), flush=True)
