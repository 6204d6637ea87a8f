#!/usr/bin/env python3
"""Time of the multigrid V-cycle (Gauss-Seidel / SpMM engine) on a 64-step slab
with 512-byte rows (ld = 64) against the same payload at the 528-byte stride the
2^J + 1 layout forces (ld = 66), and the production shape (n_loc = 65, ld = 66)."""
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
ap.add_argument('--iters', type=int, default=4)
args = ap.parse_args()
h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time)
fam = h.C_family
n_members = len(fam.members)


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


N = h.N
for n_loc, ld in ((N, N + 1), (N - 1, N - 1), (N - 1, N + 1), (N - 1, N - 1), (N, N + 1)):
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.zeros((h.M, ld), dtype=torch.float64, device='cuda')
    x[:, :n_loc] = torch.rand((h.M, n_loc), dtype=torch.float64, device='cuda', generator=g)
    out = torch.empty_like(x)
    tables = fam.slice_tables([min(k, n_members - 1) % n_members for k in range(n_loc)])
    t_k = timeit(lambda: h.Kinv_x.apply(x, out=out, n_loc=n_loc))
    t_f = timeit(lambda: fam.apply(x, out=out, n_loc=n_loc, cm=tables[0], kind=tables[1]))
    print('n_loc=%d ld=%d   Kinv_x %.3f ms (%.4f per step)   family %.3f ms (%.4f per step)'
          % (n_loc, ld, t_k, t_k / n_loc, t_f, t_f / n_loc), flush=True)
