#!/usr/bin/env python3
"""Times variants of the Kronecker-sum apply in one process, interleaved
(median of rounds): workgroup size knob and row orders."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

from bench import seeded_slab  # noqa: E402
from source import _lib  # noqa: E402
from source.assembly import space_matrices, time_matrices  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.mpi_kron import SumMPI, TridiagKronMatMPI  # noqa: E402
from source.mpi_vector import DofDistributionMPI, KronVectorMPI  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--rounds', type=int, default=7)
ap.add_argument('--orders', default='tile')
ap.add_argument('--blocks', default='0,256,512,1024')
ap.add_argument('--exp', default='0')
ap.add_argument('--modes', default='ell,csr')
ap.add_argument('--wg', default='0')
args = ap.parse_args()

mesh_space, _, mesh_time, data, _ = problem_helper('square', args.J_space, args.J_time)
A_t, L_t, M_t, G_t, u0_t = time_matrices(mesh_time)
M_x, A_x = space_matrices(mesh_space)
N, M = A_t.shape[0], M_x.shape[0]
dd = DofDistributionMPI(MPI.COMM_WORLD, N, M)
x = KronVectorMPI(dd, seeded_slab(0, N, M))
y = x._like()


def make(order):
    Mx, Ax = M_x, A_x
    if order != 'tile':
        Mx, Ax = sp.csr_matrix(M_x), sp.csr_matrix(A_x)
        if order == 'index':
            Mx.stk_row_order = Ax.stk_row_order = np.arange(M, dtype=np.int32)
    return SumMPI(dd, [TridiagKronMatMPI(dd, A_t, Mx), TridiagKronMatMPI(dd, M_t, Ax)])


from source import mpi_kron  # noqa: E402
variants = []
for mode in args.modes.split(','):
    mpi_kron._FusedKronSum.use_ell = mode == 'ell'
    for order in args.orders.split(','):
        op = make(order)
        for bs in [int(b) for b in args.blocks.split(',')]:
            for wg in [int(v) for v in args.wg.split(',')]:
                if mode == 'csr' and wg != 0:
                    continue
                variants.append((mode + '/' + order, (bs, wg), op))
nbytes = variants[0][2]._groups[0].algorithmic_bytes(N, M)
times = {(o, b): [] for o, b, _ in variants}
for rnd in range(args.rounds + 1):
    for order, bs, op in variants:
        _lib.check(_lib.lib().stk_set_tuning(b'kron_block', bs[0]))
        _lib.check(_lib.lib().stk_set_tuning(b'ell_force_generic', bs[0]))
        _lib.check(_lib.lib().stk_set_tuning(b'ell_wg_per_cu', bs[1]))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            op._matvec(x, y)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[(order, bs)].append(e0.elapsed_time(e1) / 5)
for (order, bs), t in times.items():
    med = float(np.median(t))
    print('%-10s block,wg=%-10s median %.3f ms  min %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)' %
          (order, str(bs), med, min(t), nbytes / med / 1e6, nbytes / med / 1e6 / 80))
