#!/usr/bin/env python3
"""Where the V-cycles of the two multigrid planners -- the NumPy + device planner of
source/multigrid.py and stk_mg_create_from_csr -- part: per level the Gauss-Seidel
sweeps of both plans on the same operands, then whole V-cycles with the restricted
residual fused and not."""
import ctypes
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib as stk  # noqa: E402
from source.assembly import prolongation_matrices, space_matrices  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid  # noqa: E402
from source.problem import problem_helper  # noqa: E402

problem, J_space = sys.argv[1], int(sys.argv[2])
lib = stk.lib()
if len(sys.argv) > 3:  # strip-wise smoothing forced on (strips of 1 MB), as the ctypes-only test does
    stk.check(lib.stk_set_tuning(b'mg_strip_mb', int(sys.argv[3])))
    stk.check(lib.stk_set_tuning(b'mg_strip_width', 0))


def host(m):
    m = sp.csr_matrix(m)
    m.sort_indices()
    arrs = (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float64))
    return stk.CsrHost(m.shape[0], m.shape[1], arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data), arrs


mesh = problem_helper(problem, J_space=J_space, J_time=2)[0]
M_x, A_x = space_matrices(mesh)
hier = MeshHierarchy(mesh)
P_mats = prolongation_matrices(mesh)
n_loc, ld = 7, 8
coords = np.ascontiguousarray(hier.coords, dtype=np.float64)
Ps = [host(P) for P in P_mats]
P_arr = (stk.CsrHost * len(Ps))(*[p[0] for p in Ps])
a_h, a_keep = host(A_x)
plan = ctypes.c_void_p()
stk.check(lib.stk_mg_create_from_csr(len(P_mats) + 1, ctypes.byref(a_h), None, P_arr, coords.ctypes.data,
                                     coords.shape[1], 3, 2, 1.0, 0, None, ld, ctypes.byref(plan)))
py = MultiGrid(A_x, hier, smoothsteps=3, vcycles=2)
py_plan = py._dev.ensure_plan(ld)
rng = np.random.RandomState(3)
for level in range(1, len(P_mats) + 1):
    n = py.mats[level].shape[0]
    for backward in (0, 1):
        out = []
        for handle in (plan, py_plan):
            u = torch.zeros((n, ld), dtype=torch.float64, device='cuda')
            f = torch.zeros((n, ld), dtype=torch.float64, device='cuda')
            rs = np.random.RandomState(level)
            u[:, :n_loc] = torch.from_numpy(rs.rand(n, n_loc)).cuda()
            f[:, :n_loc] = torch.from_numpy(rs.rand(n, n_loc)).cuda()
            stk.check(lib.stk_mg_smooth(handle, stk.stream(), level, n_loc, ld, 1.0, None, 3, backward,
                                        stk.ptr(f), stk.ptr(u)))
            out.append(u.clone())
        print('level %d (%d rows) %s sweeps: equal %s, max diff %.2e' % (
            level, n, 'backward' if backward else 'forward', bool(torch.equal(out[0], out[1])),
            float((out[0] - out[1]).abs().max())))
n = A_x.shape[0]
F = torch.zeros((n, ld), dtype=torch.float64, device='cuda')
F[:, :n_loc] = torch.from_numpy(rng.rand(n, n_loc)).cuda()
for opts in ({}, {'fuse_restrict': 0}):
    res = []
    for handle in (plan, py_plan):
        for k, v in opts.items():
            rc = lib.stk_mg_set_option(handle, k.encode(), v)
            if rc:
                print('   option %s not taken: %s' % (k, lib.stk_last_error()))
        u = torch.empty_like(F)
        stk.check(lib.stk_mg_apply(handle, stk.stream(), n_loc, ld, 1.0, None, None, stk.ptr(F), stk.ptr(u)))
        res.append(u.clone())
    print('V-cycles with options %s: equal %s, max rel diff %.2e' % (
        opts, bool(torch.equal(res[0], res[1])), float((res[0] - res[1]).abs().max() / res[1].abs().max())))
