"""Krylov solver of the hot loop (counterpart of reference source/linalg.py)."""
import numpy as np

from .mpi_vector import KronVectorMPI


def _zero_like(b):
    if isinstance(b, KronVectorMPI):
        return KronVectorMPI(b.dofs_distr)
    return np.zeros(b.shape)


def PCG(T, P, b, w0=None, kmax=100000, eps=1e-6, callback=None, history=None):
    """Preconditioned conjugate gradients for T w = b with preconditioner P;
    stops as soon as the algebraic error estimate r.Pr drops below eps^2
    (reference linalg.py:6-42).  Duck-typed like the reference: works on
    KronVectorMPI + LinearOperatorMPI (everything then stays on the GPU, the
    host only sees the two scalars per iteration) and on NumPy arrays.

    Returns (w, iters).  If `history` is a list, r.Pr is appended after the
    initial residual and after every iteration."""
    w = _zero_like(b) if w0 is None else w0
    record = (lambda value: None) if history is None else history.append
    threshold = eps * eps
    done = 0
    if b.dot(b) == 0:
        return w, done

    residual = b - T @ w
    direction = P @ residual
    rho = residual.dot(direction)
    record(rho)
    while rho >= threshold and done < kmax - 1:
        done += 1
        image = T @ direction
        step = rho / direction.dot(image)
        w += step * direction
        residual -= step * image
        del image  # one slab less while P works
        if callback is not None:
            callback(w, residual, done)
        z = P @ residual
        rho, previous = residual.dot(z), rho
        record(rho)
        if rho < threshold:
            break
        if hasattr(direction, 'scale_add'):  # KronVectorMPI: the same two roundings, one pass
            direction.scale_add(rho / previous, z)
        else:
            direction *= rho / previous
            direction += z
        del z
    return w, done
