"""Time-slab partition.  Test infrastructure (see oracle/__init__.py)."""
import numpy as np


def dof_distribution(N, size):
    """Block partition of N time dofs over `size` ranks: N // size each, and the
    LAST N % size ranks get one extra (reference mpi_vector.py:17-34).
    Returns the list of [t_begin, t_end) pairs."""
    assert N >= size  # mpi_vector.py:15
    block, rest = divmod(N, size)
    out, start = [], 0
    for p in range(size):
        stop = start + block + (1 if size - p - 1 < rest else 0)
        out.append((start, stop))
        start = stop
    assert start == N
    return out


def counts_displs(N, M, size):
    """Scatterv/Gatherv tables (mpi_vector.py:22-31)."""
    dist = dof_distribution(N, size)
    counts = np.array([(e - b) * M for b, e in dist], dtype=np.float64)
    displs = np.array([b * M for b, _ in dist], dtype=np.float64)
    return counts, displs


def dof2proc(N, size):
    """Owner rank of every time dof (mpi_vector.py:36-38)."""
    out = np.zeros(N)
    for p, (b, e) in enumerate(dof_distribution(N, size)):
        out[b:e] = p
    return out
