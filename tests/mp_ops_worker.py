"""Worker of tests/test_distributed.py (gpu): every operator class of
source/mpi_kron.py and the vector algebra of source/mpi_vector.py on several
ranks (sharing the test box's one GPU), each against a dense ground truth that
rank 0 forms with NumPy.  Build-owned: cases come from a table of seeded random
factors; what is compared is always  gather(op @ scatter(x))  with
dense(op) @ x.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

from oracle import wavelets as owav  # noqa: E402
from source import mpi_kron as mk  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.mpi_vector import DofDistributionMPI, KronVectorMPI  # noqa: E402
from source.wavelets import (TransposedWaveletTransformKronIdentityMPI,  # noqa: E402
                             WaveletTransformKronIdentityMPI)

TOL = 1e-12


def random_space(rng, M, density=0.2, spd=False):
    A = sp.random(M, M, density=density, random_state=rng, format='csr')
    A = A + sp.eye(M) * 2.0
    if spd:
        A = A @ A.T + sp.eye(M)
    return sp.csr_matrix(A)


def random_tridiag(rng, N):
    return sp.diags([rng.rand(N - 1), 1.0 + rng.rand(N), rng.rand(N - 1)],
                    [-1, 0, 1], format='csr')


def random_sym_pattern_time(rng, N):
    """Sparse time matrix with a symmetric pattern and long-range entries (what
    SparseKronIdentityMPI assumes of the wavelet splits)."""
    B = sp.random(N, N, density=0.15, random_state=rng, format='csr')
    B = B + B.T
    B.data[:] = rng.rand(B.nnz)
    return sp.csr_matrix(B + sp.eye(N))


def main():
    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    J = 4
    N, M = 2**J + 1, 23
    dd = DofDistributionMPI(comm, N, M)
    rng = np.random.RandomState(2024)  # the same stream on every rank

    def scattered(X):
        v = KronVectorMPI(dd)
        v.scatter(X.reshape(-1) if rank == 0 else None)
        return v

    def gathered(v):
        out = np.zeros(N * M) if rank == 0 else None
        v.gather(out)
        return out.reshape(N, M) if rank == 0 else None

    def check(name, op, dense, n_vec=2):
        """op @ x against dense @ x for seeded x, plus linearity of op."""
        for _ in range(n_vec):
            X, Y = rng.rand(N, M), rng.rand(N, M)
            a = float(rng.rand()) + 0.5
            x, y = scattered(X), scattered(Y)
            got = gathered(op @ x)
            lin = gathered(op @ (x + a * y))
            got_y = gathered(op @ y)
            if rank == 0:
                want = (dense @ X.reshape(-1)).reshape(N, M)
                err = np.linalg.norm(got - want) / np.linalg.norm(want)
                assert err < TOL, (name, err)
                err = np.linalg.norm(lin - (got + a * got_y)) / np.linalg.norm(lin)
                assert err < TOL, (name, 'linearity', err)

    T1, T2 = random_tridiag(rng, N), random_tridiag(rng, N)
    S1, S2 = random_space(rng, M), random_space(rng, M, 0.1)
    I_t, I_x = np.eye(N), np.eye(M)
    d = lambda A: A.toarray() if sp.issparse(A) else np.asarray(A)

    ops = {
        'IdentityKronMat': (mk.IdentityKronMatMPI(dd, S1), np.kron(I_t, d(S1))),
        'TridiagKronIdentity': (mk.TridiagKronIdentityMPI(dd, T1),
                                np.kron(d(T1), I_x)),
        'TridiagKronMat': (mk.TridiagKronMatMPI(dd, T1, S1),
                           np.kron(d(T1), d(S1))),
    }
    B = random_sym_pattern_time(rng, N)
    ops['SparseKronIdentity'] = (mk.SparseKronIdentityMPI(dd, B),
                                 np.kron(d(B), I_x))
    ops['SparseKronIdentity+I'] = (mk.SparseKronIdentityMPI(dd, B, True),
                                   np.kron(d(B), I_x) + np.eye(N * M))
    D = rng.rand(N, N)
    ops['MatKronIdentity'] = (mk.MatKronIdentityMPI(dd, D), np.kron(D, I_x))
    terms = [mk.TridiagKronMatMPI(dd, T1, S1), mk.TridiagKronMatMPI(dd, T2, S2),
             mk.IdentityKronMatMPI(dd, S2), mk.TridiagKronMatMPI(dd, T2, S1),
             mk.TridiagKronMatMPI(dd, T1, S2)]
    ops['Sum'] = (mk.SumMPI(dd, terms),
                  np.kron(d(T1), d(S1)) + np.kron(d(T2), d(S2)) +
                  np.kron(I_t, d(S2)) + np.kron(d(T2), d(S1)) +
                  np.kron(d(T1), d(S2)))
    ops['Composite'] = (mk.CompositeMPI(dd, [
        mk.TridiagKronMatMPI(dd, T1, S1), mk.IdentityKronMatMPI(dd, S2),
        mk.TridiagKronIdentityMPI(dd, T2)]),
        np.kron(d(T1), d(S1)) @ np.kron(I_t, d(S2)) @ np.kron(d(T2), I_x))
    blocks = [random_space(rng, M, 0.15) for _ in range(3)]
    per_t = [blocks[t % 3] for t in range(N)]
    ops['BlockDiag'] = (mk.BlockDiagMPI(dd, per_t),
                        sp.block_diag([d(b) for b in per_t]).toarray())
    Wd = owav.apply(J, np.eye(N), interleaved=True)
    ops['W'] = (WaveletTransformKronIdentityMPI(dd, J), np.kron(Wd, I_x))
    ops['WT'] = (TransposedWaveletTransformKronIdentityMPI(dd, J),
                 np.kron(Wd.T, I_x))
    for name, (op, dense) in ops.items():
        check(name, op, dense)
        assert op.num_applies > 0 and op.time_applies > 0.0, name
    for mode in ('composite', 'transpose'):
        for name in ('W', 'WT'):
            ops[name][0].mode = mode
            check(name + ':' + mode, *ops[name], n_vec=1)

    # one whole operator column by column (as_global_matrix), tiny
    dd_s = DofDistributionMPI(comm, 5, 4)
    Ts, Ss = random_tridiag(rng, 5), random_space(rng, 4, 0.5)
    G = mk.TridiagKronMatMPI(dd_s, Ts, Ss).as_global_matrix()
    if rank == 0:
        assert np.allclose(G, np.kron(d(Ts), d(Ss)), rtol=1e-13, atol=1e-14)

    # vector algebra and the all-reduced dot product
    X, Y = rng.rand(N, M), rng.rand(N, M)
    x, y = scattered(X), scattered(Y)
    assert abs(x.dot(y) - np.vdot(X, Y)) < 1e-13 * np.vdot(X, Y)
    z = 2.0 * x - y / 4.0
    z += 0.5 * y
    z *= 3.0
    got = gathered(z)
    if rank == 0:
        assert np.allclose(got, 3.0 * (2.0 * X - Y / 4.0 + 0.5 * Y), rtol=1e-15)
    c = x.copy()
    c.reset()
    assert c.dot(c) == 0.0 and abs(x.dot(x) - np.vdot(X, X)) < 1e-12 * N * M
    comm.Barrier()
    if rank == 0:
        print('mp_ops_worker ok: %d operators on %d ranks' % (len(ops), size))


if __name__ == '__main__':
    main()
