"""Operator wiring of the SERIAL heat-equation driver.
Test infrastructure (see oracle/__init__.py).

Follows reference heateq.py:18-107: trial space X = H1_t x H1_x, test space
Y = L2_t(order 1) x H1_x, B = B1 + B2 (:45-54), K = Kinv_time kron Kinv_space
(:57-63), S = B^T K B + G (:87-91), P block diagonal over the wavelet levels
(:70-85), f (:93-106), all on flat vectors through the KronLinOp formula of
reference linop.py:6-15."""
import numpy as np
import scipy.sparse as sp

from . import wavelets
from .heat import DirectInverse
from .kron import kron_linop
from .multigrid import MultiGrid


class HeatSerialOracle:
    def __init__(self, mats, J_time, precond='multigrid', smoothsteps=3,
                 vcycles=2, alpha=0.3):
        """`mats`: G_t, u0_t, Minv_Y, B1_t, B2_t, M_x, A_x, P_mats, u0_x."""
        self.__dict__.update(mats)
        self.J_time = J_time
        self.N, self.M = self.G_t.shape[0], self.M_x.shape[0]
        self.NY = self.B1_t.shape[0]
        if precond == 'multigrid':
            def mk(m):
                return MultiGrid(m, self.P_mats, smoothsteps, vcycles)
        else:
            mk = DirectInverse
        self.Kinv_x = mk(self.A_x)
        # heateq.py:66 takes WaveletTransformOp's default numbering: level by level
        self.levels = wavelets.levels(J_time, interleaved=False)
        self.C_j = [mk(sp.csr_matrix(2**j * self.M_x + alpha * self.A_x))
                    for j in range(J_time + 1)]

    @staticmethod
    def _space(op, Z):
        """op on every time row of Z (n, M)."""
        return (op @ np.ascontiguousarray(Z.T)).T

    def B(self, x):  # heateq.py:53
        return (kron_linop(self.B1_t, self.M_x, x) +
                kron_linop(self.B2_t, self.A_x, x))

    def BT(self, y):  # heateq.py:54
        return (kron_linop(sp.csr_matrix(self.B1_t.T), self.M_x, y) +
                kron_linop(sp.csr_matrix(self.B2_t.T), self.A_x, y))

    def K(self, y):  # heateq.py:63: (Kinv_time kron Kinv_space) y
        Z = self.Minv_Y @ y.reshape(self.NY, self.M)
        return self._space(self.Kinv_x, Z).reshape(-1)

    def G(self, x):  # heateq.py:55
        return kron_linop(self.G_t, self.M_x, x)

    def S(self, x):  # heateq.py:88-90
        return self.BT(self.K(self.B(x))) + self.G(x)

    def W(self, x):  # heateq.py:67
        return wavelets.apply(self.J_time, x.reshape(self.N, self.M),
                              interleaved=False).reshape(-1)

    def WT(self, x):  # heateq.py:68
        return wavelets.apply_transposed(
            self.J_time, x.reshape(self.N, self.M),
            interleaved=False).reshape(-1)

    def WT_S_W(self, x):  # heateq.py:91
        return self.WT(self.S(self.W(x)))

    def P(self, x):  # heateq.py:81-85 with linop.py:29-44
        X = x.reshape(self.N, self.M)
        out = np.empty_like(X)
        for t, j in enumerate(self.levels):
            C = self.C_j[j]
            out[t] = C @ (self.A_x @ (C @ X[t]))
        return out.reshape(-1)

    def f(self):  # heateq.py:100-106 without forcing
        return np.kron(self.u0_t, self.u0_x)
