"""Operator wiring of the parallel heat-equation driver on one rank.
Test infrastructure (see oracle/__init__.py).

Follows reference heateq_mpi.py:126-191: which matrix goes where in the Schur
complement S, the preconditioner P and the right-hand side."""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

from . import kron, wavelets
from .multigrid import MultiGrid


class DirectInverse:
    """InvLinOp (reference linop.py:18-26)."""
    def __init__(self, mat):
        self.lu = splu(sp.csc_matrix(mat),
                       options={"SymmetricMode": True},
                       permc_spec="MMD_AT_PLUS_A")

    def __matmul__(self, B):
        return self.lu.solve(np.ascontiguousarray(B))


class HeatEquationOracle:
    def __init__(self, mats, J_time, precond='multigrid', smoothsteps=3,
                 vcycles=2, alpha=0.3, use_c=True):
        """`mats`: dict with A_t, L_t, M_t, G_t, M_x, A_x, P_mats, u0_t, u0_x
        (what heateq_mpi.py:78-103 obtains from NGSolve)."""
        self.__dict__.update(mats)
        self.J_time = J_time
        self.N, self.M = self.A_t.shape[0], self.M_x.shape[0]
        self.levels = wavelets.levels(J_time, interleaved=True)
        # heateq_mpi.py:97-98
        self.Cinv_j = [
            sp.csr_matrix(2**j * self.M_x + alpha * self.A_x)
            for j in range(J_time + 1)
        ]
        if precond == 'multigrid':  # heateq_mpi.py:141-153
            mk = lambda m: MultiGrid(m, self.P_mats, smoothsteps, vcycles,
                                     use_c)
        else:  # heateq_mpi.py:154-157
            mk = DirectInverse
        self.Kinv_x = mk(self.A_x)
        self.C_j = [mk(m) for m in self.Cinv_j]

    # S = sum of 5 Kronecker terms (heateq_mpi.py:166-181)
    def S_terms(self):
        M, A, K = self.M_x, self.A_x, self.Kinv_x
        return [
            (self.A_t, [M, K, M]),
            (self.L_t, [M, K, A]),
            (sp.csr_matrix(self.L_t.T), [A, K, M]),
            (self.M_t, [A, K, A]),
            (self.G_t, [M]),
        ]

    def S(self, X):
        out = np.zeros_like(X)
        for T, ops in self.S_terms():
            Z = kron.tridiag_kron_identity(T, X)  # mpi_kron.py:215
            out += kron.composite_space(ops, Z.T).T  # mpi_kron.py:216, 149
        return out

    def W(self, X):  # wavelets.py:172-183
        return wavelets.apply(self.J_time, X, interleaved=True)

    def WT(self, X):  # wavelets.py:186-198
        return wavelets.apply_transposed(self.J_time, X, interleaved=True)

    def WT_S_W(self, X):  # heateq_mpi.py:185
        return self.WT(self.S(self.W(X)))

    def P(self, X):
        """Block diagonal of C_j A_x C_j with j = wavelet level of the time
        index (heateq_mpi.py:159-162, 183-184)."""
        out = np.empty_like(X)

        def one(t):
            C = self.C_j[self.levels[t]]
            out[t] = C @ (self.A_x @ (C @ X[t]))

        from . import multigrid
        if multigrid.THREADS > 1:  # slices are independent (fixture generation)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(multigrid.THREADS) as ex:
                list(ex.map(one, range(len(self.levels))))
        else:
            for t in range(len(self.levels)):
                one(t)
        return out

    def rhs(self):  # heateq_mpi.py:189-191
        return np.kron(self.u0_t, self.u0_x).reshape(-1, self.M)
