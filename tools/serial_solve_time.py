#!/usr/bin/env python3
"""The serial driver's solve (heateq.HeatEquation, reference heateq.py:18-158): the
reference's wiring on flat host vectors -- one PCIe round trip per operator apply --
against the same operators on device vectors (source/linop.py: DeviceLinearOperator).
    python tools/serial_solve_time.py --J_time 5 --J_space 7"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spacetime-fullgrid-parallel_amd'))
import torch  # noqa: E402
import heateq as hs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=5)
ap.add_argument('--J_space', type=int, default=7)
ap.add_argument('--precond', default='multigrid')
args = ap.parse_args()
torch.zeros(1, device='cuda')
h = hs.HeatEquation(J_space=args.J_space, J_time=args.J_time, precond=args.precond)
print('N = %d, M = %d, precond = %s' % (h.N, h.M, args.precond))
out = {}
for label, kw in (('device vectors', {}), ('host vectors', {'on_host': True})):
    h.solve(**kw)  # warm
    torch.cuda.synchronize()
    t = time.time()
    u, iters = h.solve(**kw)
    torch.cuda.synchronize()
    dt = time.time() - t
    out[label] = u
    print('%-15s %3d iterations, %8.1f ms per iteration, solve %.3f s' % (label, iters, 1e3 * dt / iters, dt))
print('solutions differ by %.2e (relative)' % (np.linalg.norm(out['device vectors'] - out['host vectors'])
                                              / np.linalg.norm(out['host vectors'])))
