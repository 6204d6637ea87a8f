"""Spatial multigrid V-cycle with Gauss-Seidel smoothing.
Test infrastructure (see oracle/__init__.py).

Follows reference source/multigrid.py:83-97 (smoother), :130-197 (MultiGrid).
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# Fixture generation at BASELINE sizes (tests/golden/make_oracle_vectors.py)
# sets this > 1: the right-hand sides of a batch are independent, so a batch is
# cut into row chunks that run side by side.  Every right-hand side still sees
# exactly the sequential arithmetic above; the result does not depend on it.
THREADS = 1


def build_c():
    """Compile oracle/gs.c (called by __graft_entry__.build and lazily here)."""
    so = os.path.join(_HERE, 'liboracle_gs.so')
    src = os.path.join(_HERE, 'gs.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle_gs.so'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c())
        i32p = np.ctypeslib.ndpointer(np.int32, flags='C')
        f64p = np.ctypeslib.ndpointer(np.float64, flags='C')
        _LIB.oracle_gs_sweeps.argtypes = [
            ctypes.c_int32, i32p, i32p, f64p, f64p, f64p, f64p, ctypes.c_int32,
            ctypes.c_int32
        ]
        _LIB.oracle_gs_sweeps.restype = None
        _LIB.oracle_gs_sweeps_batch.argtypes = [
            ctypes.c_int32, i32p, i32p, f64p, f64p, f64p, f64p, ctypes.c_int32,
            ctypes.c_int32, ctypes.c_int32
        ]
        _LIB.oracle_gs_sweeps_batch.restype = None
    return _LIB


def gs_python(mat, u, f, its, backward):
    """Literal restatement of Smoother.PreSmooth / PostSmooth
    (multigrid.py:89-97), `its` times.  Pure Python: small cases only."""
    mat = sp.csr_matrix(mat)
    invdiag = mat.diagonal()**-1
    ip, ix, dat = mat.indptr, mat.indices, mat.data
    n = mat.shape[0]
    order = range(n - 1, -1, -1) if backward else range(n)
    for _ in range(its):
        for i in order:
            ax = 0.0
            for k in range(ip[i], ip[i + 1]):
                ax += dat[k] * u[ix[k]]
            u[i] += invdiag[i] * (f[i] - ax)
    return u


class Smoother:
    """`its` Gauss-Seidel sweeps per call, forward (PreSmooth) or backward
    (PostSmooth), in dof order (multigrid.py:83-97, 116-127)."""
    def __init__(self, mat, its, use_c=True):
        mat = sp.csr_matrix(mat)
        mat.sort_indices()
        self.n = mat.shape[0]
        self.indptr = np.ascontiguousarray(mat.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(mat.indices, dtype=np.int32)
        self.data = np.ascontiguousarray(mat.data, dtype=np.float64)
        self.invdiag = np.ascontiguousarray(mat.diagonal()**-1)
        self.mat = mat
        self.its = its
        self.use_c = use_c

    def _sweep(self, u, f, backward):
        assert u.flags.c_contiguous and u.dtype == np.float64
        f = np.ascontiguousarray(f, dtype=np.float64)
        if not self.use_c:
            gs_python(self.mat, u, f, self.its, backward)
        elif u.ndim == 1:
            _lib().oracle_gs_sweeps(self.n, self.indptr, self.indices,
                                    self.data, self.invdiag, u, f, self.its,
                                    int(backward))
        else:  # (k, n): one right-hand side per row
            _lib().oracle_gs_sweeps_batch(self.n, self.indptr, self.indices,
                                          self.data, self.invdiag, u, f,
                                          u.shape[0], self.its, int(backward))

    def PreSmooth(self, u, f):
        self._sweep(u, f, False)

    def PostSmooth(self, u, f):
        self._sweep(u, f, True)


def galerkin_hierarchy(mat, P_mats):
    """mats[j] = R_j mats[j+1] P_j, coarse to fine (multigrid.py:142-145)."""
    mats = [sp.csr_matrix(mat)]
    for P in reversed(P_mats):
        mats.insert(0, sp.csr_matrix(P.T.tocsr() @ mats[0] @ P))
    return mats


class MultiGrid:
    """`vcycles` V-cycles from a zero initial guess (multigrid.py:130-197).

    ``apply(b)`` takes one right-hand side (M,) or a batch (k, M) with one
    right-hand side per row (the reference's default _matmat loops over the
    columns and calls _matvec, which is the same thing)."""
    def __init__(self, mat, P_mats, smoothsteps=2, vcycles=1, use_c=True):
        self.P_mats = [sp.csr_matrix(P) for P in P_mats]
        self.R_mats = [P.T.tocsr() for P in self.P_mats]
        self.J = len(P_mats)
        self.smoothsteps = smoothsteps
        self.vcycles = vcycles
        self.mats = galerkin_hierarchy(mat, self.P_mats)
        self.smoothers = [None] + [
            Smoother(self.mats[j], smoothsteps, use_c)
            for j in range(1, self.J + 1)
        ]
        # coarse solve (multigrid.py:161-165)
        self.coarse_solver = splu(sp.csc_matrix(self.mats[0].T),
                                  options={"SymmetricMode": True},
                                  permc_spec="MMD_AT_PLUS_A")
        self.shape = self.mats[-1].shape
        import threading
        self._coarse_lock = threading.Lock()

    def MGM(self, j, u_j, f_j):
        """multigrid.py:168-182; u_j, f_j of shape (n_j,) or (k, n_j)."""
        if j == 0:
            with self._coarse_lock:  # SuperLU solve objects are not re-entrant
                u_j[...] = self.coarse_solver.solve(
                    np.ascontiguousarray(f_j.T)).T
            return
        self.smoothers[j].PreSmooth(u_j, f_j)
        A, R, P = self.mats[j], self.R_mats[j - 1], self.P_mats[j - 1]
        d_c = np.ascontiguousarray((R @ (A @ u_j.T - f_j.T)).T)
        u_c = np.zeros_like(d_c)
        self.MGM(j - 1, u_c, d_c)
        u_j -= (P @ u_c.T).T
        self.smoothers[j].PostSmooth(u_j, f_j)

    def apply(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        if THREADS > 1 and b.ndim == 2 and b.shape[0] > 1:
            from concurrent.futures import ThreadPoolExecutor
            cuts = np.linspace(0, b.shape[0], min(THREADS, b.shape[0]) + 1).astype(int)

            def chunk(k):
                xs, bs = x[cuts[k]:cuts[k + 1]], b[cuts[k]:cuts[k + 1]]
                for _ in range(self.vcycles):
                    self.MGM(self.J, xs, bs)

            with ThreadPoolExecutor(len(cuts) - 1) as ex:
                list(ex.map(chunk, range(len(cuts) - 1)))
            return x
        for _ in range(self.vcycles):
            self.MGM(self.J, x, b)
        return x

    def __matmul__(self, B):
        """Space-operator convention of the reference: acts on (M,) vectors or
        (M, k) blocks with one right-hand side per COLUMN (linop.py:75-79)."""
        B = np.asarray(B)
        if B.ndim == 1:
            return self.apply(B)
        return self.apply(np.ascontiguousarray(B.T)).T
