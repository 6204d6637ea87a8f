#!/usr/bin/env python3
"""InvLinOp above the dense limit (reference linop.py:18-26): SuperLU's factors applied on
the device (stk_lu_solve: level-scheduled triangular solves, csrc/sptrsv.hip) against the
round trip through SuperLU on the host of rounds 1-5, per apply and in the whole solve of
BASELINE config 1 with precond='direct'.

    python tools/direct_solve_time.py [--J_space 6] [--J_time 3]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source.linalg import PCG  # noqa: E402
from source.linop import InvLinOp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--J_space', type=int, default=6)
    ap.add_argument('--J_time', type=int, default=3)
    args = ap.parse_args()
    t = time.time()
    h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time, precond='direct')
    print('set-up %.2f s: N = %d, M = %d' % (time.time() - t, h.N, h.M))
    K = h.Kinv_x
    print('K^-1: %d / %d dependency levels of the L / U solve, %d launches per apply' % K.levels())
    x = h.rhs.buf
    n_loc = h.rhs.n_loc
    for host in (False, True):
        K.host_solve = host
        K.apply(x, n_loc=n_loc)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10):
            K.apply(x, n_loc=n_loc)
        torch.cuda.synchronize()
        print('K^-1 apply, %s: %.2f ms' % ('SuperLU on the host + PCIe both ways' if host else 'device', (time.time() - t) * 100))
    for host in (False, True):
        InvLinOp.host_solve = host
        for op in [h.Kinv_x] + list(h.C_j):
            op.host_solve = host
        hist = []
        torch.cuda.synchronize()
        t = time.time()
        w, it = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
        torch.cuda.synchronize()
        dt = time.time() - t
        print('solve, %s: %d iterations, %.1f ms per iteration, last r.Pr %.3e' % (
            'host' if host else 'device', it, dt / max(it, 1) * 1e3, hist[-1]))
    InvLinOp.host_solve = False


if __name__ == '__main__':
    main()
