"""Build-owned P1 assembly, replacing the NGSolve wrappers of the reference
(reference source/ngsolve_helper.py:21-94).

Conventions follow NGSolve / the reference:
* ``mat[i, j] = a(phi_j, phi_i)`` (trial function = column, test = row);
* matrices are restricted to free dofs and explicit zeros are dropped
  (ngsolve_helper.py:38-45);
* dtypes are float64 data, int32 indices/indptr (mpi_shared_mem.py:46-48).
"""
import numpy as np
import scipy.sparse as sp


def _finish(mat):
    mat = sp.csr_matrix(mat)
    mat.sum_duplicates()
    mat.eliminate_zeros()
    mat.sort_indices()
    mat = sp.csr_matrix((mat.data.astype(np.float64),
                         mat.indices.astype(np.int32),
                         mat.indptr.astype(np.int32)),
                        shape=mat.shape)
    return mat


# ----------------------------------------------------------------------------
# Time: P1 on the uniform interval, no Dirichlet rows (heateq_mpi.py:78-88).
# ----------------------------------------------------------------------------
def time_matrices(mesh_time):
    """Returns A_t (stiffness), L_t (u * v'), M_t (mass), G_t (trace at the
    start), and the load vector u0_t of v(0) (heateq_mpi.py:78-88, 100-101)."""
    n = mesh_time.nv
    h = mesh_time.h
    ne = n - 1
    i = np.arange(ne)
    rows = np.concatenate([i, i, i + 1, i + 1])
    cols = np.concatenate([i, i + 1, i, i + 1])

    def elementwise(k00, k01, k10, k11):
        vals = np.concatenate([
            np.full(ne, k00), np.full(ne, k01), np.full(ne, k10),
            np.full(ne, k11)
        ])
        return _finish(sp.coo_matrix((vals, (rows, cols)), shape=(n, n)))

    A_t = elementwise(1 / h, -1 / h, -1 / h, 1 / h)
    M_t = elementwise(h / 3, h / 6, h / 6, h / 3)
    # L_t[i, j] = int phi_j phi_i'  (trial u = phi_j, test gradient)
    L_t = elementwise(-0.5, -0.5, 0.5, 0.5)
    G_t = _finish(sp.coo_matrix(([1.0], ([0], [0])), shape=(n, n)))
    u0_t = np.zeros(n)
    u0_t[0] = 1.0
    return A_t, L_t, M_t, G_t, u0_t


# ----------------------------------------------------------------------------
# Space: P1 on a triangulation.
# ----------------------------------------------------------------------------
def _tri_geometry(mesh):
    p = mesh.points
    t = mesh.tris
    x0, x1, x2 = p[t[:, 0]], p[t[:, 1]], p[t[:, 2]]
    e1 = x1 - x0
    e2 = x2 - x0
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    area = 0.5 * np.abs(det)
    # gradients of the barycentric coordinates
    g = np.empty((len(t), 3, 2))
    g[:, 1, 0] = e2[:, 1] / det
    g[:, 1, 1] = -e2[:, 0] / det
    g[:, 2, 0] = -e1[:, 1] / det
    g[:, 2, 1] = e1[:, 0] / det
    g[:, 0] = -g[:, 1] - g[:, 2]
    return area, g


def free_dofs(mesh):
    return np.flatnonzero(~mesh.boundary)


def _restrict(mat, fd):
    return _finish(sp.csr_matrix(mat)[fd, :].tocsc()[:, fd].tocsr())


def tile_row_order(mesh, rows_per_tile=None):
    """A processing order for the free dofs that follows the geometry: the
    bounding box is cut into square tiles of about `rows_per_tile` vertices,
    tiles are visited row by row and vertices inside a tile lexicographically.
    The hierarchical dof numbering scatters the neighbours of a vertex over
    the whole index range; visiting rows in this order instead keeps the
    gathers of consecutive workgroups inside the same few hundred KB, i.e. in
    the L2 of the XCD that runs them.  Purely a performance hint
    (stk_kron_sum_apply's row_ids); results do not depend on it."""
    import os
    if rows_per_tile is None:
        rows_per_tile = int(os.environ.get('STK_ROWS_PER_TILE', '2048'))
    fd = free_dofs(mesh)
    p = mesh.points[fd]
    lo, hi = p.min(axis=0), p.max(axis=0)
    ext = np.maximum(hi - lo, 1e-300)
    ntiles = max(1.0, len(fd) / float(rows_per_tile))
    side = np.sqrt(ext[0] * ext[1] / ntiles)
    tx = np.floor((p[:, 0] - lo[0]) / side).astype(np.int64)
    ty = np.floor((p[:, 1] - lo[1]) / side).astype(np.int64)
    order = np.lexsort((p[:, 0], p[:, 1], tx, ty))
    return order.astype(np.int32)


def space_matrices(mesh):
    """Mass M_x and stiffness A_x on the free dofs (heateq_mpi.py:91-96)."""
    area, g = _tri_geometry(mesh)
    t = mesh.tris
    nv = mesh.nv
    rows = np.repeat(t, 3, axis=1).reshape(-1)
    cols = np.tile(t, (1, 3)).reshape(-1)
    K = np.einsum('tid,tjd->tij', g, g) * area[:, None, None]
    Mloc = (np.ones((3, 3)) + np.eye(3)) / 12.0
    Mv = area[:, None, None] * Mloc[None]
    A = sp.coo_matrix((K.reshape(-1), (rows, cols)), shape=(nv, nv)).tocsr()
    M = sp.coo_matrix((Mv.reshape(-1), (rows, cols)), shape=(nv, nv)).tocsr()
    # the three-direction mesh yields exact zeros on the diagonal edges only up
    # to rounding; drop what is numerically zero, as eliminate_zeros would for
    # NGSolve's exactly integrated entries
    A.data[np.abs(A.data) < 1e-14 * np.abs(A.data).max()] = 0.0
    fd = free_dofs(mesh)
    M, A = _restrict(M, fd), _restrict(A, fd)
    M.stk_row_order = A.stk_row_order = tile_row_order(mesh)
    return M, A


# Dunavant degree-4 rule (6 points)
_QW = np.array([0.223381589678011] * 3 + [0.109951743655322] * 3)
_a, _b = 0.445948490915965, 0.108103018168070
_c, _d = 0.091576213509771, 0.816847572980459
_QL = np.array([[_b, _a, _a], [_a, _b, _a], [_a, _a, _b], [_d, _c, _c],
                [_c, _d, _c], [_c, _c, _d]])


def space_load(mesh, fn):
    """int fn * phi_i on the free dofs (heateq_mpi.py:102-103)."""
    area, _ = _tri_geometry(mesh)
    p = mesh.points
    t = mesh.tris
    X = np.einsum('ql,tld->tqd', _QL, p[t])  # quadrature points
    f = fn(X[..., 0], X[..., 1])  # (nt, nq)
    loc = np.einsum('tq,q,ql->tl', f, _QW, _QL) * area[:, None]
    vec = np.zeros(mesh.nv)
    np.add.at(vec, t.reshape(-1), loc.reshape(-1))
    return vec[free_dofs(mesh)]


def prolongation_matrices(mesh):
    """P_mats[j]: free dofs of level j -> free dofs of level j+1, built from the
    parent-vertex table exactly as reference multigrid.py:39-59:
    identity on the old vertices, 1/2 + 1/2 at the two parents of a new one."""
    fd_mask = ~mesh.boundary
    P_mats = []
    for j in range(mesh.J):
        nc, nf = mesh.nverts[j], mesh.nverts[j + 1]
        nnew = nf - nc
        row = np.concatenate([np.arange(nc), np.repeat(np.arange(nc, nf), 2)])
        col = np.concatenate([np.arange(nc), mesh.parents[nc:nf].reshape(-1)])
        val = np.concatenate([np.ones(nc), np.full(2 * nnew, 0.5)])
        P = sp.csr_matrix((val, (row, col)), shape=(nf, nc))
        fr = np.flatnonzero(fd_mask[:nf])
        fc = np.flatnonzero(fd_mask[:nc])
        P_mats.append(_finish(P[fr, :].tocsc()[:, fc].tocsr()))
    # drop levels without any free dof (the 2-triangle square has none)
    while P_mats and P_mats[0].shape[1] == 0:
        P_mats.pop(0)
    return P_mats
