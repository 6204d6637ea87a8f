"""Build-owned P1 assembly, replacing the NGSolve wrappers of the reference
(reference source/ngsolve_helper.py:21-94).

Conventions follow NGSolve / the reference:
* ``mat[i, j] = a(phi_j, phi_i)`` (trial function = column, test = row);
* matrices are restricted to free dofs and explicit zeros are dropped
  (ngsolve_helper.py:38-45);
* dtypes are float64 data, int32 indices/indptr (mpi_shared_mem.py:46-48).
"""
import os

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


def time_matrices_test_space(mesh_time):
    """Time factors of the serial driver (reference heateq.py:37-63), whose test
    space Y is L2(order 1) in time: discontinuous P1, two dofs per element
    (2e, 2e + 1 = the two nodal functions of element e).  Returns
      M_Y   (2Ne x 2Ne)  mass matrix of Y_t (block diagonal),
      Minv_Y             its inverse (`Preconditioner(A_bf.time.bf, 'direct')`),
      B1_t  (2Ne x N)    int phi_j' psi_i   (B1_bf, heateq.py:45-46),
      B2_t  (2Ne x N)    int phi_j  psi_i   (B2_bf, heateq.py:47-48).
    NGSolve uses a Legendre basis for L2; S = B^T K B does not depend on the
    basis of Y because K inverts the mass matrix of Y exactly."""
    n, h = mesh_time.nv, mesh_time.h
    ne = n - 1
    e = np.arange(ne)
    r = np.concatenate([2 * e, 2 * e, 2 * e + 1, 2 * e + 1])

    def blocks(k00, k01, k10, k11, cols):
        vals = np.concatenate([np.full(ne, k00), np.full(ne, k01),
                               np.full(ne, k10), np.full(ne, k11)])
        shape = (2 * ne, 2 * ne if cols is r_cols else n)
        return _finish(sp.coo_matrix((vals, (r, cols)), shape=shape))

    r_cols = np.concatenate([2 * e, 2 * e + 1, 2 * e, 2 * e + 1])
    x_cols = np.concatenate([e, e + 1, e, e + 1])
    M_Y = blocks(h / 3, h / 6, h / 6, h / 3, r_cols)
    Minv_Y = blocks(4 / h, -2 / h, -2 / h, 4 / h, r_cols)
    B1_t = blocks(-0.5, 0.5, -0.5, 0.5, x_cols)
    B2_t = blocks(h / 3, h / 6, h / 6, h / 3, x_cols)
    return M_Y, Minv_Y, B1_t, B2_t


# ----------------------------------------------------------------------------
# Space: P1 on a triangulation (d = 2) or a tetrahedral mesh (d = 3).
# ----------------------------------------------------------------------------
def _simplex_geometry(mesh):
    """Volumes and the gradients of the barycentric coordinates per cell
    (closed-form inverse of the d x d edge matrix; cached on the mesh)."""
    cached = getattr(mesh, '_stk_geometry', None)
    if cached is not None and cached[0] == len(mesh.cells):
        return cached[1], cached[2]
    p = mesh.points
    c = mesh.cells
    d = c.shape[1] - 1
    E = p[c[:, 1:]] - p[c[:, :1]]  # (nc, d, d): rows = edge vectors
    adj = np.empty_like(E)  # adjugate: E^{-1} = adj / det
    if d == 2:
        adj[:, 0, 0], adj[:, 0, 1] = E[:, 1, 1], -E[:, 0, 1]
        adj[:, 1, 0], adj[:, 1, 1] = -E[:, 1, 0], E[:, 0, 0]
        det = E[:, 0, 0] * E[:, 1, 1] - E[:, 0, 1] * E[:, 1, 0]
    else:
        a, b, c3 = (E[:, 0, k] for k in range(3))  # components of the rows, by column
        d_, e_, f_ = (E[:, 1, k] for k in range(3))
        g_, h_, i_ = (E[:, 2, k] for k in range(3))
        # cofactors written out (np.cross on (n, 3) arrays is several times slower)
        adj[:, 0, 0], adj[:, 1, 0], adj[:, 2, 0] = e_ * i_ - f_ * h_, f_ * g_ - d_ * i_, d_ * h_ - e_ * g_
        adj[:, 0, 1], adj[:, 1, 1], adj[:, 2, 1] = c3 * h_ - b * i_, a * i_ - c3 * g_, b * g_ - a * h_
        adj[:, 0, 2], adj[:, 1, 2], adj[:, 2, 2] = b * f_ - c3 * e_, c3 * d_ - a * f_, a * e_ - b * d_
        det = a * adj[:, 0, 0] + b * adj[:, 1, 0] + c3 * adj[:, 2, 0]
    vol = np.abs(det) / (2.0 if d == 2 else 6.0)
    g = np.empty((len(c), d + 1, d))
    # column k of E^{-1} is the gradient of lambda_{k+1}
    g[:, 1:] = np.swapaxes(adj, 1, 2) / det[:, None, None]
    g[:, 0] = -g[:, 1:].sum(axis=1)
    mesh._stk_geometry = (len(c), vol, g)
    return vol, g


def free_dofs(mesh):
    return np.flatnonzero(~mesh.boundary)


def _restrict(mat, fd):
    """Rows and columns `fd` (ascending) of a matrix, explicit zeros dropped
    (ngsolve_helper.py:38-45): a filter over the CSR arrays -- the entries keep
    their values and their order, so the result is that of
    `mat[fd, :][:, fd]` bit for bit, without the three format conversions."""
    mat = sp.csr_matrix(mat)
    mat.sum_duplicates()
    n = mat.shape[0]
    free = np.zeros(n, dtype=bool)
    free[fd] = True
    new_id = np.cumsum(free) - 1
    rows_of = np.repeat(np.arange(n), np.diff(mat.indptr))
    keep = free[rows_of] & free[mat.indices] & (mat.data != 0)
    counts = np.bincount(new_id[rows_of[keep]], minlength=len(fd))
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    out = sp.csr_matrix((mat.data[keep].astype(np.float64),
                         new_id[mat.indices[keep]].astype(np.int32), indptr),
                        shape=(len(fd), len(fd)))
    out.has_sorted_indices = True
    return out


def _rows_per_tile(rows_per_tile):
    import os
    if rows_per_tile is None:
        rows_per_tile = int(os.environ.get('STK_ROWS_PER_TILE', '2048'))
    return rows_per_tile


# point sets of at least this many dofs are ordered by libstk's host threads
TILE_ORDER_ON_HOST_THREADS = 16384


def tile_order_from_coords(coords, rows_per_tile=None, small_lexsort=True):
    """A processing order that follows the geometry: the bounding box is cut
    into square (cubic) tiles of about `rows_per_tile` vertices, tiles are
    visited lexicographically and so are the vertices inside a tile."""
    rows_per_tile = _rows_per_tile(rows_per_tile)
    p = np.asarray(coords)
    d = p.shape[1]
    lex = tuple(p[:, k] for k in range(d))
    if small_lexsort and len(p) <= rows_per_tile:
        return np.lexsort(lex).astype(np.int32)
    lo, hi = p.min(axis=0), p.max(axis=0)
    ext = np.maximum(hi - lo, 1e-30)
    ntiles = max(1.0, len(p) / float(rows_per_tile))
    side = (np.prod(ext) / ntiles)**(1.0 / d)
    snake = os.environ.get('STK_TILE_WALK') == 'snake' and d == 2
    if len(p) >= TILE_ORDER_ON_HOST_THREADS and d in (2, 3) and not snake and __package__:
        # the same order on the host threads of libstk (stk_tile_order: a sample sort;
        # tests/test_host_cpu.py test_plan_helpers_on_host_threads).  Not when this file
        # is loaded alone, without its package (the fixture generator under tests/golden)
        from . import _lib
        pts = np.ascontiguousarray(p, dtype=np.float64)
        corner = np.ascontiguousarray(lo, dtype=np.float64)
        order = np.empty(len(p), dtype=np.int32)
        _lib.check(_lib.lib().stk_tile_order(len(p), d, pts.ctypes.data, corner.ctypes.data, float(side),
                                             order.ctypes.data))
        return order
    tiles = tuple(np.floor((p[:, k] - lo[k]) / side).astype(np.int64)
                  for k in range(d))
    if snake:
        # EXPERIMENT (measured and not adopted):
        # a boustrophedon walk inside a tile -- every other mesh row of a tile from right to
        # left, so that the two readers of a gathered row above / below sit closer together
        ys = np.unique(p[:, 1])
        row = np.searchsorted(ys, p[:, 1])
        lex = (np.where(row & 1, -p[:, 0], p[:, 0]), p[:, 1])
    return np.lexsort(lex + tiles).astype(np.int32)


def tile_rows_from_coords(coords, rows_per_tile=None):
    """Index of the tile ROW (the slowest key of tile_order_from_coords) every
    point falls into; all zero when the points fit one tile."""
    rows_per_tile = _rows_per_tile(rows_per_tile)
    p = np.asarray(coords)
    d = p.shape[1]
    if len(p) <= rows_per_tile:
        return np.zeros(len(p), dtype=np.int32)
    lo, hi = p.min(axis=0), p.max(axis=0)
    ext = np.maximum(hi - lo, 1e-30)
    ntiles = max(1.0, len(p) / float(rows_per_tile))
    side = (np.prod(ext) / ntiles)**(1.0 / d)
    return np.floor((p[:, d - 1] - lo[d - 1]) / side).astype(np.int32)


def tile_row_order(mesh, rows_per_tile=None):
    """A processing order for the free dofs that follows the geometry: the
    bounding box is cut into square tiles of about `rows_per_tile` vertices,
    tiles are visited row by row and vertices inside a tile lexicographically.
    The hierarchical dof numbering scatters the neighbours of a vertex over
    the whole index range; visiting rows in this order instead keeps the
    gathers of consecutive workgroups inside the same few hundred KB, i.e. in
    the L2 of the XCD that runs them.  Purely a performance hint
    (stk_kron_sum_apply's row_ids); results do not depend on it."""
    return tile_order_from_coords(mesh.points[free_dofs(mesh)], rows_per_tile,
                                  small_lexsort=False)


def _space_matrices_libstk(mesh):
    """The 2-D assembly on the host threads of libstk (csrc/assemble.hip,
    stk_p1_assemble_2d): rows summed triangle by triangle in ascending triangle
    number.  Bit for bit the matrices of the SciPy path below on uniformly
    refined meshes (every partial sum there is exact or a sum of equal terms)."""
    import ctypes

    from . import _lib
    lib = _lib.lib()
    pts = np.ascontiguousarray(mesh.points, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.cells, dtype=np.int64)
    bnd = np.ascontiguousarray(mesh.boundary, dtype=np.uint8)
    res = ctypes.c_void_p()
    _lib.check(lib.stk_p1_assemble_2d(mesh.nv, len(cells), pts.ctypes.data, cells.ctypes.data,
                                      bnd.ctypes.data, 1e-14, ctypes.byref(res)))
    try:
        n_free, nnz_a, nnz_m = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.stk_p1_result_sizes(res, ctypes.byref(n_free), ctypes.byref(nnz_a),
                                           ctypes.byref(nnz_m)))
        mats = []
        for which, nnz in ((0, nnz_a.value), (1, nnz_m.value)):
            indptr = np.empty(n_free.value + 1, dtype=np.int32)
            indices = np.empty(nnz, dtype=np.int32)
            data = np.empty(nnz, dtype=np.float64)
            _lib.check(lib.stk_p1_result_copy(res, which, indptr.ctypes.data, indices.ctypes.data,
                                              data.ctypes.data))
            mat = sp.csr_matrix((data, indices, indptr), shape=(n_free.value, n_free.value))
            mat.has_sorted_indices = True
            mat.has_canonical_format = True
            mats.append(mat)
    finally:
        lib.stk_p1_result_free(res)
    return mats[1], mats[0]


def space_matrices(mesh, scipy_path=False):
    """Mass M_x and stiffness A_x on the free dofs (heateq_mpi.py:91-96).
    Triangulations are assembled by libstk (stk_p1_assemble_2d); tetrahedral
    meshes, and scipy_path=True, take the NumPy / SciPy form below."""
    if mesh.cells.shape[1] == 3 and not scipy_path:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            order = pool.submit(tile_row_order, mesh)
            M, A = _space_matrices_libstk(mesh)
            order = order.result()
        M.stk_row_order = A.stk_row_order = order
        return M, A
    vol, g = _simplex_geometry(mesh)
    c = mesh.cells
    nl = c.shape[1]
    nv = mesh.nv
    rows = np.repeat(c, nl, axis=1).reshape(-1)
    cols = np.tile(c, (1, nl)).reshape(-1)
    K = np.matmul(g, np.swapaxes(g, 1, 2)) * vol[:, None, None]
    Mloc = (np.ones((nl, nl)) + np.eye(nl)) / (nl * (nl + 1.0))
    Mv = vol[:, None, None] * Mloc[None]
    fd = free_dofs(mesh)

    def stiffness():
        A = sp.coo_matrix((K.reshape(-1), (rows, cols)), shape=(nv, nv)).tocsr()
        # the three-direction mesh (Kuhn mesh in 3-D) yields exact zeros on the
        # diagonal edges only up to rounding; drop what is numerically zero, as
        # eliminate_zeros would for NGSolve's exactly integrated entries
        A.data[np.abs(A.data) < 1e-14 * np.abs(A.data).max()] = 0.0
        return _restrict(A, fd)

    def mass():
        M = sp.coo_matrix((Mv.reshape(-1), (rows, cols)), shape=(nv, nv)).tocsr()
        return _restrict(M, fd)

    # two independent sparse assemblies and the processing order: side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=3) as pool:
        A, M, order = (pool.submit(stiffness), pool.submit(mass),
                       pool.submit(tile_row_order, mesh))
        A, M, order = A.result(), M.result(), order.result()
    M.stk_row_order = A.stk_row_order = order
    return M, A


# Dunavant degree-4 rule on the triangle (6 points)
_QW = np.array([0.223381589678011] * 3 + [0.109951743655322] * 3)
_a, _b = 0.445948490915965, 0.108103018168070
_c, _d = 0.091576213509771, 0.816847572980459
_QL = np.array([[_b, _a, _a], [_a, _b, _a], [_a, _a, _b], [_d, _c, _c],
                [_c, _d, _c], [_c, _c, _d]])


def _keast4():
    """Keast's degree-4 rule on the tetrahedron (11 points, weights sum to 1)."""
    import itertools
    pts, wts = [[0.25] * 4], [-0.0789333333333333]
    a, b = 0.0714285714285714, 0.785714285714286
    for k in range(4):
        q = [a] * 4
        q[k] = b
        pts.append(q)
        wts.append(0.0457333333333333)
    a, b = 0.399403576166799, 0.100596423833201
    for i, j in itertools.combinations(range(4), 2):
        q = [b] * 4
        q[i] = q[j] = a
        pts.append(q)
        wts.append(0.149333333333333)
    return np.array(wts), np.array(pts)


_QW3, _QL3 = _keast4()


def _simplex_volumes(mesh):
    """The volumes alone (same expressions as _simplex_geometry)."""
    c = mesh.cells
    if c.shape[1] != 3 or getattr(mesh, '_stk_geometry', None) is not None:
        return _simplex_geometry(mesh)[0]
    p = mesh.points
    p0 = p[c[:, 0]]
    e0, e1 = p[c[:, 1]] - p0, p[c[:, 2]] - p0
    return np.abs(e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0]) / 2.0


def _space_load_libstk(mesh, fn):
    """The 2-D load vector around libstk's host threads (csrc/mesh_refine.hip): the
    quadrature points and the sums there, fn -- a pointwise function of arrays, as a
    coefficient function is -- evaluated here, in slices side by side (NumPy's
    elementwise loops release the interpreter lock)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    from . import _lib
    lib = _lib.lib()
    pts = np.ascontiguousarray(mesh.points, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.cells, dtype=np.int64)
    nt, nq = len(cells), len(_QW)
    qw, ql = np.ascontiguousarray(_QW), np.ascontiguousarray(_QL)
    qx, qy = np.empty((nt, nq)), np.empty((nt, nq))
    _lib.check(lib.stk_p1_load_points_2d(mesh.nv, nt, pts.ctypes.data, cells.ctypes.data, nq, ql.ctypes.data,
                                         qx.ctypes.data, qy.ctypes.data))
    f = np.empty((nt, nq))
    parts = max(1, min(8, os.cpu_count() or 1, nt // 16384))
    cuts = [nt * k // parts for k in range(parts + 1)]

    def evaluate(k):
        f[cuts[k]:cuts[k + 1]] = fn(qx[cuts[k]:cuts[k + 1]], qy[cuts[k]:cuts[k + 1]])

    if parts == 1:
        evaluate(0)
    else:
        with ThreadPoolExecutor(max_workers=parts) as pool:
            list(pool.map(evaluate, range(parts)))
    vec = np.empty(mesh.nv)
    _lib.check(lib.stk_p1_load_sum_2d(mesh.nv, nt, pts.ctypes.data, cells.ctypes.data, nq, qw.ctypes.data,
                                      ql.ctypes.data, f.ctypes.data, vec.ctypes.data))
    return vec[free_dofs(mesh)]


def space_load(mesh, fn, numpy_path=False):
    """int fn * phi_i on the free dofs (heateq_mpi.py:102-103).  Triangulations: on
    the host threads of libstk (stk_p1_load_points_2d / stk_p1_load_sum_2d; the same
    sums in a fixed order, within rounding of the NumPy form below, which tetrahedral
    meshes and numpy_path=True take)."""
    if mesh.cells.shape[1] == 3 and not numpy_path:
        return _space_load_libstk(mesh, fn)
    vol = _simplex_volumes(mesh)
    p = mesh.points
    c = mesh.cells
    qw, ql = (_QW, _QL) if c.shape[1] == 3 else (_QW3, _QL3)
    X = np.matmul(ql, p[c])  # quadrature points (nt, nq, d)
    f = fn(*(X[..., k] for k in range(X.shape[-1])))  # (nt, nq)
    loc = np.matmul(f * qw, ql) * vol[:, None]  # (nt, nl)
    vec = np.bincount(c.reshape(-1), weights=loc.reshape(-1), minlength=mesh.nv)
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
