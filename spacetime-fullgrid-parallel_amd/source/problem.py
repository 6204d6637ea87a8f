"""Model problems (counterpart of reference source/problem.py:7-41).

``square``: u(t,x,y) = exp(-2 pi^2 t) sin(pi x) sin(pi y) on [0,1]^2, no forcing
(problem.py:7-19).  ``lshape`` is not in the reference (anything but square /
cube asserts there, problem.py:35-41); BASELINE.json config 4 names it, so it
is defined here with the same data on the L-shaped domain.  ``cube``:
u = exp(-3 pi^2 t) sin(pi x) sin(pi y) sin(pi z) on [0,1]^3 (problem.py:21-32).
"""
import numpy as np

from .mesh import (construct_2d_lshape_mesh, construct_2d_square_mesh,
                   construct_3d_cube_mesh, construct_interval)


def _time_mesh(J_space, J_time):
    if not J_time:
        J_time = J_space
    return construct_interval(N=2**int(J_time + 0.5))


def _u0(x, y):
    return np.sin(np.pi * x) * np.sin(np.pi * y)


def square(J_space, J_time=None):
    mesh_space, bc = construct_2d_square_mesh(nrefines=J_space)
    data = {'g': [], 'u0': _u0}
    return mesh_space, bc, _time_mesh(J_space, J_time), data, "square"


def lshape(J_space, J_time=None):
    mesh_space, bc = construct_2d_lshape_mesh(nrefines=J_space)
    data = {'g': [], 'u0': _u0}
    return mesh_space, bc, _time_mesh(J_space, J_time), data, "lshape"


def jitter(mesh, rel=0.2, seed=0):
    """Moves every interior vertex by up to `rel` times the shortest edge (seeded):
    the triangulation keeps its topology and hierarchy, but no two elements are
    congruent any more, so no two entries of M_x or A_x repeat -- an "unstructured
    mesh, irregular CSR" in the sense of BASELINE.json config 4 for the kernels
    that otherwise live on repeated values (dictionary form of the Kronecker
    apply)."""
    pts, tris = mesh.points, mesh.tris
    e = np.concatenate([pts[tris[:, a]] - pts[tris[:, b]] for a, b in ((0, 1), (1, 2), (2, 0))])
    h = np.sqrt((e * e).sum(axis=1)).min()
    rng = np.random.RandomState(seed)
    move = rel * h * (2.0 * rng.rand(*pts.shape) - 1.0)
    move[mesh.boundary] = 0.0
    mesh.points = pts + move
    return mesh


def lshape_jitter(J_space, J_time=None):
    mesh_space, bc, mesh_time, data, _ = lshape(J_space, J_time)
    return jitter(mesh_space), bc, mesh_time, data, "lshape_jitter"


def _u0_3d(x, y, z):
    return np.sin(np.pi * x) * np.sin(np.pi * y) * np.sin(np.pi * z)


def cube(J_space, J_time=None):
    mesh_space, bc = construct_3d_cube_mesh(nrefines=J_space)
    data = {'g': [], 'u0': _u0_3d}
    return mesh_space, bc, _time_mesh(J_space, J_time), data, "cube"


def problem_helper(problem, J_space, J_time=None):
    if problem == 'square':
        return square(J_space, J_time)
    elif problem == 'lshape':
        return lshape(J_space, J_time)
    elif problem == 'lshape_jitter':
        return lshape_jitter(J_space, J_time)
    elif problem == 'cube':
        return cube(J_space, J_time)
    else:
        assert (False)
