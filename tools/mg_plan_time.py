#!/usr/bin/env python3
"""Host time of the multigrid plan construction at the bench configuration: the
NumPy / SciPy planner of source/multigrid.py against stk_mg_create_from_csr
(host C++ inside libstk), for K = MultiGrid(A_x) and for the preconditioner
family 2^j M_x + alpha A_x."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib as stk  # noqa: E402
from source.assembly import prolongation_matrices, space_matrices  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--J_time', type=int, default=6)
args = ap.parse_args()
torch.zeros(1, device='cuda')
lib = stk.lib()
mesh = problem_helper('square', J_space=args.J_space, J_time=args.J_time)[0]
M_x, A_x = space_matrices(mesh)
hier = MeshHierarchy(mesh)
P_mats = prolongation_matrices(mesh)
ld = 2**args.J_time + 2


def host(m):
    m = sp.csr_matrix(m)
    m.sort_indices()
    arrs = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float64))
    return stk.CsrHost(m.shape[0], m.shape[1], arrs[0].ctypes.data, arrs[1].ctypes.data,
                       arrs[2].ctypes.data), arrs


coords = np.ascontiguousarray(hier.coords, dtype=np.float64)
Ps = [host(P) for P in P_mats]
P_arr = (stk.CsrHost * len(Ps))(*[p[0] for p in Ps])
a_h, a_keep = host(A_x)
m_h, m_keep = host(M_x)
cms = np.array([2.0**j for j in range(args.J_time + 1)])
for rep in range(2):
    t = time.time()
    plan = ctypes.c_void_p()
    stk.check(lib.stk_mg_create_from_csr(len(P_mats) + 1, ctypes.byref(a_h), None, P_arr, coords.ctypes.data,
                                         coords.shape[1], 3, 2, 1.0, 0, None, ld, ctypes.byref(plan)))
    t_k = time.time() - t
    t = time.time()
    fam = ctypes.c_void_p()
    stk.check(lib.stk_mg_create_from_csr(len(P_mats) + 1, ctypes.byref(a_h), ctypes.byref(m_h), P_arr,
                                         coords.ctypes.data, coords.shape[1], 3, 2, 0.3, len(cms),
                                         cms.ctypes.data, ld, ctypes.byref(fam)))
    t_f = time.time() - t
    stk.check(lib.stk_mg_destroy(plan))
    stk.check(lib.stk_mg_destroy(fam))
    print('libstk planner:  K %.2f s   family %.2f s' % (t_k, t_f), flush=True)
t = time.time()
MultiGrid(A_x, hier, smoothsteps=3, vcycles=2)
t_k = time.time() - t
t = time.time()
MultiGridFamily(A_x, M_x, hier, ca=0.3, cms=list(cms), smoothsteps=3, vcycles=2)
print('NumPy planner:   K %.2f s   family %.2f s' % (t_k, time.time() - t), flush=True)
