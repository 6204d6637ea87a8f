"""Kronecker-product operator applies on the full (N, M) array.
Test infrastructure (see oracle/__init__.py).

All functions take and return arrays of shape (N, M), time-major, which is
``KronVectorMPI.X_loc`` for a single rank (reference mpi_vector.py:62-71).
"""
import numpy as np
import scipy.sparse as sp


def identity_kron_mat(mat_space, X):
    """(I_t kron X_x) x = (X_x @ X^T)^T  (reference mpi_kron.py:143-150)."""
    return np.ascontiguousarray((mat_space @ X.T).T)


def tridiag_kron_identity(mat_time, X):
    """(T_t kron I_x) x = T_t @ X  (mpi_kron.py:186-201; the halo slicing of
    :165-183 is the identity on one rank)."""
    return mat_time @ X


def tridiag_kron_mat(mat_time, mat_space, X):
    """(T_t kron X_x) x: time factor first, then the space factor on the
    result (mpi_kron.py:214-219)."""
    return identity_kron_mat(mat_space, tridiag_kron_identity(mat_time, X))


def sum_apply(terms, X):
    """SumMPI (mpi_kron.py:77-90): sum_k (T_k kron X_k) x, accumulated in the
    order of the list.  `terms` = [(mat_time, mat_space), ...]."""
    out = np.zeros_like(X)
    for T, S in terms:
        out += tridiag_kron_mat(T, S, X)
    return out


def kron_linop(mat_time, mat_space, x):
    """Serial KronLinOp (linop.py:6-15) on the flat vector."""
    K, L = mat_time.shape[1], mat_space.shape[1]
    return (mat_space @ (mat_time @ x.reshape(K, L)).T).T.reshape(-1)


def composite_space(linops, X):
    """CompositeLinOp (linop.py:68-79): right-to-left application of space
    operators on (M, k) blocks."""
    Y = X
    for op in reversed(linops):
        Y = op @ Y
    return Y


def block_diag(space_ops, X):
    """BlockDiagMPI (mpi_kron.py:122-132): y[t] = op_t @ x[t]."""
    out = np.empty_like(X)
    for t, op in enumerate(space_ops):
        out[t] = op @ X[t]
    return out


def sparse_kron_identity(mat_time, X, add_identity=False):
    """SparseKronIdentityMPI (mpi_kron.py:285-317): loop over the COO triplets
    of the time matrix, one AXPY of length M each, on top of a copy of the
    input (add_identity) or of zero."""
    coo = sp.coo_matrix(mat_time)
    out = X.copy() if add_identity else np.zeros_like(X)
    for t, idx, c in zip(coo.row, coo.col, coo.data):
        out[t, :] += c * X[idx]
    return out


def dense_kron(mat_time, mat_space):
    """Ground truth used by the reference tests (mpi_kron.py:221-222)."""
    T = mat_time.toarray() if sp.issparse(mat_time) else np.asarray(mat_time)
    S = mat_space.toarray() if sp.issparse(mat_space) else np.asarray(mat_space)
    return np.kron(T, S)
