#!/usr/bin/env python3
"""Serial driver: Andreev's method on tensor-product trial spaces, through the
serial LinearOperator surface of source/linop.py (counterpart of reference
heateq.py:18-158).

Same structure as the reference: X = H1_t x H1_x, Y = L2_t(order 1) x H1_x,
B = B1 + B2, K = Kinv_time kron Kinv_space, S = B^T K B + G, P block diagonal
over the wavelet levels, solved with PCG.  The matrices come from the build's own
P1 assembly (source/assembly.py) instead of NGSolve; every operator application
runs on the GPU.  The operators accept the reference's flat NumPy vectors (one
round trip over PCIe per apply) and device vectors (source/linop.py:
DeviceLinearOperator); solve() keeps the whole iteration on the device.  The
time-parallel path is heateq_mpi.py."""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source.assembly import (space_load, space_matrices,  # noqa: E402
                             time_matrices, time_matrices_test_space)
from source.linalg import PCG  # noqa: E402
from source.linop import (BlockDiagLinOp, CompositeLinOp,  # noqa: E402
                          DeviceLinearOperator, InvLinOp, KronLinOp,
                          device_vector, host_vector)
from source.multigrid import MeshHierarchy, MultiGrid  # noqa: E402
from source.problem import problem_helper  # noqa: E402
from source.wavelets import WaveletTransformOp  # noqa: E402


class HeatEquation:
    """Implementation of Andreev's method for tensor-product trial spaces
    (reference heateq.py:18-107)."""
    def __init__(self, J_space=2, J_time=None, problem='square',
                 precond='multigrid', alpha=0.3, smoothsteps=3, vcycles=2):
        if J_time is None:
            J_time = J_space
        mesh_space, bc, mesh_time, data, fn = problem_helper(problem,
                                                             J_space=J_space,
                                                             J_time=J_time)
        A_t, L_t, M_t, G_t, u0_t = time_matrices(mesh_time)
        M_Y, Minv_Y, B1_t, B2_t = time_matrices_test_space(mesh_time)
        M_x, A_x = space_matrices(mesh_space)
        self.N, self.M = A_t.shape[0], M_x.shape[0]
        self.N_Y = M_Y.shape[0]  # time dofs of the test space
        self.M_x, self.A_x = M_x, A_x
        self.time_mats = dict(A_t=A_t, L_t=L_t, M_t=M_t, G_t=G_t, u0_t=u0_t,
                              Minv_Y=Minv_Y, B1_t=B1_t, B2_t=B2_t)

        # B = B1 + B2 (heateq.py:45-54), G (:49-50, :55)
        self.B = KronLinOp(B1_t, M_x) + KronLinOp(B2_t, A_x)
        self.BT = (KronLinOp(sp.csr_matrix(B1_t.T), M_x) +
                   KronLinOp(sp.csr_matrix(B2_t.T), A_x))
        self.G = KronLinOp(G_t, M_x)

        if precond == 'multigrid':
            self.hierarchy = MeshHierarchy(mesh_space)

            def mk(mat):
                return MultiGrid(mat, self.hierarchy, smoothsteps=smoothsteps,
                                 vcycles=vcycles)
        else:
            self.hierarchy = None
            mk = InvLinOp
        # preconditioner on Y (heateq.py:57-63)
        self.K = KronLinOp(Minv_Y, mk(A_x))

        # wavelet transform (heateq.py:65-68)
        W_t = WaveletTransformOp(J_time)
        eye = sp.eye(self.M, format='csr')
        self.W = KronLinOp(W_t, eye)
        self.WT = KronLinOp(W_t.T, eye)

        # preconditioner on X (heateq.py:70-85)
        self.alpha = alpha
        self.C_j = [mk(sp.csr_matrix(2**j * M_x + alpha * A_x))
                    for j in range(J_time + 1)]
        self.CAC_j = [CompositeLinOp([self.C_j[j], A_x, self.C_j[j]])
                      for j in range(J_time + 1)]
        self.P = BlockDiagLinOp([self.CAC_j[j] for j in W_t.levels])

        # Schur complement (heateq.py:87-91)
        # (a DeviceLinearOperator: the same expression maps flat host vectors, as the
        # reference's LinearOperator does, and device vectors)
        self.S = DeviceLinearOperator(
            self.G.shape,
            matvec=lambda v: self.BT @ (self.K @ (self.B @ v)) + self.G @ v)
        self.WT_S_W = self.WT @ self.S @ self.W

        # right-hand side (heateq.py:93-106); the model problems have no
        # forcing (data['g'] is empty)
        assert not data['g'], 'forcing terms are not wired'
        self.g_vec = np.zeros(self.K.shape[0])
        self.u0_x = space_load(mesh_space, data['u0'])
        self.f = self.BT @ (self.K @ self.g_vec) + np.kron(u0_t, self.u0_x)

    def solve(self, callback=None, on_host=False):
        """PCG on the wavelet-transformed system; returns (u, iterations).  The
        right-hand side goes to the device once and the solution comes back once: in
        between, every vector of the iteration is device-resident (the callback sees
        device vectors; source.linop.host_vector copies one out).  on_host=True runs
        the reference's wiring literally -- flat NumPy vectors, one round trip over
        PCIe per operator apply."""
        if on_host:
            w, iters = PCG(self.WT_S_W, self.P, self.WT @ self.f, callback=callback)
            return self.W @ w, iters
        rhs = device_vector(self.f, self.N)
        w, iters = PCG(self.WT_S_W, self.P, self.WT @ rhs, callback=callback)
        return host_vector(self.W @ w), iters

    def errors(self, u):
        """(algebraic error of u in the X-norm, error in Y') as the reference's
        driver reports them (heateq.py:147-153)."""
        u = device_vector(u, self.N)
        residual = device_vector(self.f, self.N) - self.S @ u
        defect = device_vector(self.g_vec, self.N_Y) - self.B @ u
        return residual.dot(self.P @ residual), defect.dot(self.K @ defect)


_OPTIONS = (
    ('problem', str, 'square', 'problem type (square, lshape, cube)'),
    ('J_time', int, 5, 'number of time refines'),
    ('J_space', int, 6, 'number of space refines'),
    ('precond', str, 'multigrid', 'spatial preconditioner: multigrid or direct.'),
    ('alpha', float, 0.3, 'Alpha value used in the preconditioner for X.'),
)


def main(argv=None):
    parser = argparse.ArgumentParser(description='Solve the heat equation, serial wiring.')
    for flag, kind, default, text in _OPTIONS:
        parser.add_argument('--' + flag, type=kind, default=default, help=text)
    args = parser.parse_args(argv)
    print('Arguments: %s' % args)
    print('\n\nCreating HeatEquation with %d time refines and %d space refines.'
          % (args.J_time, args.J_space))
    heat = HeatEquation(**vars(args))
    print('Size of time mesh: %d dofs. Size of space mesh: %d dofs' % (heat.N, heat.M))
    print('Solving: ', end='')
    u, iters = heat.solve(callback=lambda w, residual, k: print('.', end='', flush=True))
    print('Done in %d  PCG steps. X-norm algebraic error: %s. Error in Yprime: %s\n'
          % ((iters,) + heat.errors(u)))
    return heat, u, iters


if __name__ == '__main__':
    main()
