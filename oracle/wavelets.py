"""3-point wavelet -> hat-function transform in time.
Test infrastructure (see oracle/__init__.py).

Follows reference source/wavelets.py:45-198.  Two independent restatements
are kept so they can check each other: explicit sparse matrices p(j), q(j),
split(j), and a stencil form that is also the specification of the GPU kernel.
"""
import numpy as np
import scipy.sparse as sp


def prolong(j):
    """p(j): hats of level j-1 -> hats of level j, shape (2^j+1, 2^(j-1)+1)
    (wavelets.py:81-90): 1 at even nodes, 1/2 + 1/2 at odd ones."""
    nc, nf = 2**(j - 1) + 1, 2**j + 1
    r = np.concatenate([2 * np.arange(nc), 2 * np.arange(nc - 1) + 1,
                        2 * np.arange(nc - 1) + 1])
    c = np.concatenate([np.arange(nc), np.arange(nc - 1), np.arange(1, nc)])
    v = np.concatenate([np.ones(nc), np.full(2 * (nc - 1), 0.5)])
    return sp.csr_matrix((v, (r, c)), shape=(nf, nc))


def embed(j):
    """q(j): 3-point wavelets of level j -> hats of level j, shape
    (2^j+1, 2^(j-1)) (wavelets.py:92-104): 2^(j/2) * (-1/2, 1, -1/2), with -1
    instead of -1/2 at the two end nodes."""
    if j == 0:
        return sp.identity(2, format='csr')
    nw, nf = 2**(j - 1), 2**j + 1
    m = np.arange(nw)
    r = np.concatenate([2 * m, 2 * m + 1, 2 * m + 2])
    c = np.concatenate([m, m, m])
    v = np.concatenate([np.full(nw, -0.5), np.ones(nw), np.full(nw, -0.5)])
    q = sp.lil_matrix(sp.csr_matrix((v, (r, c)), shape=(nf, nw)))
    q[0, 0] = -1
    q[nf - 1, nw - 1] = -1
    return sp.csr_matrix(q) * 2**(j / 2)


def levels(J, interleaved=True):
    """Level of the wavelet attached to every index (wavelets.py:70-79)."""
    if interleaved:
        lv = np.zeros(2**J + 1, dtype=int)
        for j in reversed(range(J + 1)):
            lv[::2**(J - j)] = j
        return lv
    counts = [2] + [2**(j - 1) for j in range(1, J + 1)]
    return np.array([j for j, n in enumerate(counts) for _ in range(n)])


def apply(J, X, interleaved=True):
    """W @ X, X of shape (2^J+1, k) (wavelets.py:106-118)."""
    Y = np.array(X, dtype=np.float64, copy=True)
    for j in range(1, J + 1):
        p, q = prolong(j), embed(j)
        if interleaved:
            S = 2**(J - j)
            Y[::S] = p @ Y[::S][::2] + q @ Y[::S][1::2]
        else:
            nc, nf = 2**(j - 1) + 1, 2**j + 1
            Y[:nf] = p @ Y[:nc] + q @ Y[nc:nf]
    return Y


def apply_transposed(J, X, interleaved=True):
    """W^T @ X (wavelets.py:120-134)."""
    Y = np.array(X, dtype=np.float64, copy=True)
    for j in reversed(range(1, J + 1)):
        pT, qT = prolong(j).T.tocsr(), embed(j).T.tocsr()
        if interleaved:
            S = 2**(J - j)
            fine = Y[::S].copy()
            Y[::S][::2] = pT @ fine
            Y[::S][1::2] = qT @ fine
        else:
            nc, nf = 2**(j - 1) + 1, 2**j + 1
            fine = Y[:nf].copy()
            Y[:nc] = pT @ fine
            Y[nc:nf] = qT @ fine
    return Y


def split(J, j):
    """[p(j) q(j)] - I on the stride-2^(J-j) nodes, interleaved numbering
    (wavelets.py:136-169).  W = prod_j (I + split(j)), coarse level first."""
    n = 2**J + 1
    if j == 0:
        return sp.csr_matrix((n, n))
    S = 2**(J - j)
    p, q = prolong(j).tocoo(), embed(j).tocoo()
    rows = np.concatenate([S * p.row, S * q.row, np.arange(0, n, S)])
    cols = np.concatenate([2 * S * p.col, S + 2 * S * q.col,
                           np.arange(0, n, S)])
    vals = np.concatenate([p.data, q.data, -np.ones(len(range(0, n, S)))])
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, n))


# ---------------------------------------------------------------------------
# Stencil form (the specification the HIP kernel implements).
# ---------------------------------------------------------------------------
def level_step(J, j, X):
    """One level of W in the interleaved numbering as a 3-point stencil on the
    stride-S nodes, S = 2^(J-j), reading only pre-level values:
      odd  k: y = 1/2 (x[(k-1)S] + x[(k+1)S]) + s x[kS]
      even k: y = x[kS] - 1/2 s (x[(k-1)S] + x[(k+1)S]),  ends: - s x[(k+-1)S]
    with s = 2^(j/2)."""
    S, s = 2**(J - j), 2**(j / 2)
    Y = X.copy()
    n = 2**j
    for k in range(n + 1):
        if k % 2 == 1:
            Y[k * S] = 0.5 * (X[(k - 1) * S] + X[(k + 1) * S]) + s * X[k * S]
        elif k == 0:
            Y[0] = X[0] - s * X[S]
        elif k == n:
            Y[n * S] = X[n * S] - s * X[(n - 1) * S]
        else:
            Y[k * S] = X[k * S] - 0.5 * s * (X[(k - 1) * S] + X[(k + 1) * S])
    return Y


def level_step_transposed(J, j, X):
    """One level of W^T (the transpose of level_step):
      even k: y = x[kS] + 1/2 (x[(k-1)S] + x[(k+1)S])   (missing ends dropped)
      odd  k: y = s (x[kS] - c_l x[(k-1)S] - c_r x[(k+1)S]),
              c = 1 if the neighbour is an end node (0 or 2^J) else 1/2."""
    S, s = 2**(J - j), 2**(j / 2)
    Y = X.copy()
    n = 2**j
    for k in range(n + 1):
        if k % 2 == 0:
            acc = X[k * S].copy()
            if k > 0:
                acc = acc + 0.5 * X[(k - 1) * S]
            if k < n:
                acc = acc + 0.5 * X[(k + 1) * S]
            Y[k * S] = acc
        else:
            cl = 1.0 if k - 1 == 0 else 0.5
            cr = 1.0 if k + 1 == n else 0.5
            Y[k * S] = s * (X[k * S] - cl * X[(k - 1) * S] -
                            cr * X[(k + 1) * S])
    return Y


def apply_stencil(J, X):
    Y = np.array(X, dtype=np.float64, copy=True)
    for j in range(1, J + 1):
        Y = level_step(J, j, Y)
    return Y


def apply_transposed_stencil(J, X):
    Y = np.array(X, dtype=np.float64, copy=True)
    for j in reversed(range(1, J + 1)):
        Y = level_step_transposed(J, j, Y)
    return Y
