"""Krylov solver of the hot loop (counterpart of reference source/linalg.py)."""
import numpy as np

from .mpi_vector import KronVectorMPI


def PCG(T, P, b, w0=None, kmax=100000, eps=1e-6, callback=None, history=None):
    """Preconditioned conjugate gradients for T w = b with preconditioner P;
    stops as soon as the algebraic error estimate r.Pr drops below eps^2
    (reference linalg.py:6-42).  Duck-typed like the reference: works on
    KronVectorMPI + LinearOperatorMPI (everything then stays on the GPU, the
    host only sees the two scalars per iteration) and on NumPy arrays.

    Returns (w, iters).  If `history` is a list, r.Pr is appended after the
    initial residual and after every iteration."""
    if w0 is not None:
        w = w0
    elif isinstance(b, KronVectorMPI):
        w = KronVectorMPI(b.dofs_distr)
    else:
        w = np.zeros(b.shape)

    iters = 0
    if b.dot(b) == 0:
        return w, iters

    r = b - T @ w
    p = P @ r
    rho = r.dot(p)
    if history is not None:
        history.append(rho)
    if rho < eps * eps:
        return w, iters

    for k in range(1, kmax):
        iters += 1
        Tp = T @ p
        step = rho / p.dot(Tp)
        w += step * p
        r -= step * Tp
        del Tp
        if callback is not None:
            callback(w, r, k)
        z = P @ r
        rho_prev, rho = rho, r.dot(z)
        if history is not None:
            history.append(rho)
        if rho < eps * eps:
            break
        p *= rho / rho_prev
        p += z
        del z
    return w, iters
