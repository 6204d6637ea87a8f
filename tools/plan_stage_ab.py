#!/usr/bin/env python3
"""The plans stage of the set-up (K's multigrid plan, the preconditioner family's, the
Kronecker plan, side by side as HeatEquationMPI builds them) with either multigrid plan
from the NumPy + device planner (holds the interpreter lock) or from
stk_mg_create_from_csr (host threads of libstk, no lock)."""
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source.host_malloc import keep_to_the_heap  # noqa: E402
keep_to_the_heap()  # the drivers' allocator policy (STK_KEEP_MALLOC=1: untouched, as in round 5)

import numpy as np  # noqa: E402
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib as stk  # noqa: E402
from source.assembly import prolongation_matrices, space_matrices  # noqa: E402
from source.linop import EllMatrices, forget_union_pattern  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily  # noqa: E402
from source.problem import problem_helper  # noqa: E402

J_space, J_time = 9, 6
torch.zeros(1, device='cuda')
lib = stk.lib()
mesh = problem_helper('square', J_space=J_space, J_time=J_time)[0]
M_x, A_x = space_matrices(mesh)
P_mats = prolongation_matrices(mesh)
ld = 2**J_time + 2
cms = np.array([2.0**j for j in range(J_time + 1)])


def host(m):
    m = sp.csr_matrix(m)
    m.sort_indices()
    arrs = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float64))
    return stk.CsrHost(m.shape[0], m.shape[1], arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data), arrs


def c_plan(hier, family):
    coords = np.ascontiguousarray(hier.coords, dtype=np.float64)
    Ps = [host(P) for P in P_mats]
    P_arr = (stk.CsrHost * len(Ps))(*[p[0] for p in Ps])
    a_h, a_keep = host(A_x)
    m_h, m_keep = host(M_x)
    plan = ctypes.c_void_p()
    if family:
        stk.check(lib.stk_mg_create_from_csr(len(P_mats) + 1, ctypes.byref(a_h), ctypes.byref(m_h), P_arr,
                                             coords.ctypes.data, coords.shape[1], 3, 2, 0.3, len(cms),
                                             cms.ctypes.data, ld, ctypes.byref(plan)))
    else:
        stk.check(lib.stk_mg_create_from_csr(len(P_mats) + 1, ctypes.byref(a_h), None, P_arr, coords.ctypes.data,
                                             coords.shape[1], 3, 2, 1.0, 0, None, ld, ctypes.byref(plan)))
    return plan


def stage(k_c, fam_c):
    A, Mx = sp.csr_matrix(A_x.copy()), sp.csr_matrix(M_x.copy())  # fresh arrays: no plan cache hits
    hier = MeshHierarchy(mesh).prepare()
    forget_union_pattern()
    on = stk.in_device_context
    t = time.time()
    with ThreadPoolExecutor(max_workers=4) as pool:
        jobs = [pool.submit(on(lambda: EllMatrices([Mx, A]).packed_for(65)))]
        jobs.append(pool.submit(on(c_plan), hier, False) if k_c else
                    pool.submit(on(MultiGrid), A, hier, smoothsteps=3, vcycles=2, gs_rows='owned'))
        jobs.append(pool.submit(on(c_plan), hier, True) if fam_c else
                    pool.submit(on(MultiGridFamily), A, Mx, hier, ca=0.3, cms=list(cms), smoothsteps=3, vcycles=2,
                                gs_rows='owned'))
        out = [j.result() for j in jobs]
    torch.cuda.synchronize()
    dt = time.time() - t
    for o in out[1:]:
        if isinstance(o, ctypes.c_void_p):
            stk.check(lib.stk_mg_destroy(o))
    return dt


for rep in range(3):
    print('plans stage: Python K + Python family %.2f s | C K + Python family %.2f s | Python K + C family %.2f s | '
          'C K + C family %.2f s' % (stage(0, 0), stage(1, 0), stage(0, 1), stage(1, 1)), flush=True)
