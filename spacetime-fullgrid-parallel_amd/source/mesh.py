"""Build-owned meshes: uniform interval in time, red-refined triangulations in space.

Replaces the reference's Netgen/NGSolve meshing (reference source/mesh.py:4-43,
source/problem.py:7-41), which cannot be installed here.  Only the properties
the hot path relies on are kept:

* the time mesh is the uniform interval [0, T] with N = 2^J_time + 1 nodes
  (reference mesh.py:4-18, problem.py:13-14);
* the space mesh is a conforming triangulation, uniformly (red) refined
  ``nrefines`` times, and vertices are numbered *hierarchically*: the vertices
  of level l are a prefix of the vertices of level l+1, which is what
  MeshHierarchy relies on (reference multigrid.py:20-32, 53-55).

Within one refinement level the numbering is free (Netgen's own order is not
knowable here).  We use that freedom for the GPU: the new vertices of a level
are grouped by *edge colour* (no two edges of one triangle share a colour, and
red refinement propagates a valid colouring to the children), and sorted
lexicographically inside a colour.  New vertices of one colour are then never
adjacent, old vertices are never adjacent to each other in the refined mesh,
and therefore a Gauss-Seidel sweep in dof order (reference multigrid.py:89-97)
has a dependency DAG of depth (#colours + 1) -- 4 for the square -- while still
being exactly the sequential sweep of the reference.  The colour of the
hypotenuses comes LAST by default: the stiffness matrix has no entry across a
hypotenuse, so the midpoints of the two short-edge classes share a dependency
level and its sweep has 3.  HYPOTENUSE_FIRST = True puts that colour first, which
leaves the sweep with 2 levels (a red-black split) -- cheaper applies, but a
weaker smoother and a longer solve: measured and not the default (see the flag).
"""
import numpy as np

# Where the hypotenuse class goes among a level's new vertices.  False (the default):
# LAST -- the stiffness matrix sweeps in 3 dependency groups, [old | short-edge
# midpoints | hypotenuse midpoints].  True: FIRST -- 2 groups, the red-black split
# [old, hypotenuse midpoints | short-edge midpoints], 5-13 % cheaper applies of S.
# The sweep runs in dof order (reference multigrid.py:89-97), so this is a choice of
# SMOOTHER, and the red-black order is the weaker one: kappa(K^-1 A_x) 1.028 against
# 1.002 at J_space = 9 (profiles/r04_mg_sweep_J9.log), PCG needs 14 iterations instead
# of 13 at J_time = 6 / J_space = 9 and 12 instead of 9 at J_time = 3, and the time to
# solution is 5-23 % LONGER on every configuration measured
# (profiles/r04_numbering_*.log; tools/numbering_ab.py, tools/mg_sweep.py).  The
# fixtures under tests/golden hold the default.
HYPOTENUSE_FIRST = False
# tools/numbering_ab.py --orders: colour c of the coarsest mesh's edges becomes class
# CLASS_ORDER[c] (None: as coloured; with the default colouring of a three-direction
# mesh, colour 0 / 1 / 2 = horizontal / vertical / hypotenuse).
CLASS_ORDER = None
# True: TriangleMesh.refine takes its NumPy form (the fixture generator under
# tests/golden loads this file alone, without libstk); the tables are the same.
REFINE_NUMPY = False


class IntervalMesh:
    """Uniform mesh of [0, T] with N_el elements (reference mesh.py:4-18)."""
    def __init__(self, N_el=16, T=1.0):
        self.N_el = int(N_el)
        self.T = float(T)
        self.nodes = self.T * np.arange(self.N_el + 1) / self.N_el
        self.h = self.T / self.N_el

    @property
    def nv(self):
        return self.N_el + 1


class TriangleMesh:
    """A hierarchy of red-refined triangulations with hierarchical numbering.

    Attributes (finest level unless stated):
      points (nv, 2), tris (nt, 3), boundary (nv,) bool,
      nverts[l]   -- number of vertices of level l (prefix sizes),
      parents (nv, 2) -- parent vertex pair of every vertex (-1 on level 0),
                          the analogue of NGSolve's GetParentVertices
                          (reference multigrid.py:20-21),
      vcolor (nv,) -- colour class of the edge a vertex bisected (-1 on level 0).
    """
    def __init__(self, points, tris, boundary_fn):
        self.points = np.asarray(points, dtype=np.float64)
        self.tris = np.asarray(tris, dtype=np.int64)
        self.boundary_fn = boundary_fn
        nv = len(self.points)
        self.nverts = [nv]
        self.parents = -np.ones((nv, 2), dtype=np.int64)
        self.vcolor = -np.ones(nv, dtype=np.int64)
        self._tri_edge_color = self._greedy_edge_colouring()
        self.boundary = boundary_fn(self.points)

    # ------------------------------------------------------------------
    def _edges(self):
        """Unique edges, and for every triangle the ids of its 3 edges
        (edge k is opposite to local vertex k)."""
        t = self.tris
        e = np.stack([t[:, [1, 2]], t[:, [2, 0]], t[:, [0, 1]]], axis=1)
        e = np.sort(e.reshape(-1, 2), axis=1)
        nv = len(self.points)
        key = e[:, 0] * nv + e[:, 1]
        ukey, inv = np.unique(key, return_inverse=True)
        edges = np.stack([ukey // nv, ukey % nv], axis=1)
        return edges, inv.reshape(-1, 3)

    def _greedy_edge_colouring(self):
        """Colour the edges of the (tiny) coarsest mesh so that the 3 edges of
        every triangle get 3 different colours.  Returns per-triangle colours
        (nt, 3) aligned with the local edge numbering."""
        edges, te = self._edges()
        ne = len(edges)
        # edges conflict when they share a triangle
        nbrs = [set() for _ in range(ne)]
        for tri in te:
            for a in tri:
                for b in tri:
                    if a != b:
                        nbrs[a].add(b)
        # process edges grouped by length, then direction: structured meshes get the
        # natural (horizontal, vertical, diagonal) classes.  The stiffness matrix of
        # a right triangle has no entry across its hypotenuse, so A_x couples neither
        # two old vertices, nor an old vertex with a hypotenuse midpoint, nor two
        # short-edge midpoints.  Longest edges LAST (the default): the dof-order sweep
        # over A_x has 3 dependency levels, [old | short-edge midpoints | hypotenuse
        # midpoints].  Longest edges FIRST (HYPOTENUSE_FIRST): {old vertices,
        # hypotenuse midpoints} and {short-edge midpoints} are a red-black split that
        # respects the hierarchical prefix, 2 levels.  M_x + A_x has 4 either way.
        d = self.points[edges[:, 1]] - self.points[edges[:, 0]]
        ang = np.round(np.mod(np.arctan2(d[:, 1], d[:, 0]), np.pi), 9)
        length = np.round(np.hypot(d[:, 0], d[:, 1]), 9)
        order = np.lexsort((np.arange(ne), ang, -length if HYPOTENUSE_FIRST else length))
        col = -np.ones(ne, dtype=np.int64)
        for e in order:
            used = {col[n] for n in nbrs[e] if col[n] >= 0}
            c = 0
            while c in used:
                c += 1
            col[e] = c
        if CLASS_ORDER is not None:  # experiments: another order of the colour classes
            col = np.asarray(CLASS_ORDER, dtype=np.int64)[col]
        return col[te]

    # ------------------------------------------------------------------
    def refine(self, numpy_path=None):
        """One uniform red refinement; new vertices are appended.  On the host threads
        of libstk (stk_tri_refine, csrc/mesh_refine.hip); numpy_path=True takes the
        NumPy form below -- the same tables entry for entry
        (tests/test_host_cpu.py test_refinement_on_host_threads_matches_numpy)."""
        if not (REFINE_NUMPY if numpy_path is None else numpy_path):
            return self._refine_libstk()
        edges, te = self._edges()
        ne = len(edges)
        nv = len(self.points)
        # colour per unique edge (consistent by construction)
        ecol = np.empty(ne, dtype=np.int64)
        ecol[te.reshape(-1)] = self._tri_edge_color.reshape(-1)
        mid = 0.5 * (self.points[edges[:, 0]] + self.points[edges[:, 1]])
        # numbering of the new vertices: by colour, then y, then x
        order = np.lexsort((mid[:, 0], mid[:, 1], ecol))
        rank = np.empty(ne, dtype=np.int64)
        rank[order] = np.arange(ne)
        new_id = nv + rank

        self.points = np.vstack([self.points, mid[order]])
        self.parents = np.vstack([self.parents, edges[order]])
        self.vcolor = np.concatenate([self.vcolor, ecol[order]])
        self.nverts.append(nv + ne)

        t = self.tris
        m = new_id[te]  # m[:, k] = midpoint of the edge opposite vertex k
        c = self._tri_edge_color
        v0, v1, v2 = t[:, 0], t[:, 1], t[:, 2]
        m0, m1, m2 = m[:, 0], m[:, 1], m[:, 2]
        c0, c1, c2 = c[:, 0], c[:, 1], c[:, 2]
        # children; edge k of a child is opposite its local vertex k and carries
        # the colour of the parent edge it is parallel to / a half of.
        tris = np.concatenate([
            np.stack([v0, m2, m1], 1),
            np.stack([v1, m0, m2], 1),
            np.stack([v2, m1, m0], 1),
            np.stack([m0, m1, m2], 1),
        ])
        cols = np.concatenate([
            np.stack([c0, c1, c2], 1),
            np.stack([c1, c2, c0], 1),
            np.stack([c2, c0, c1], 1),
            np.stack([c0, c1, c2], 1),
        ])
        self.tris = tris
        self._tri_edge_color = cols
        self.boundary = self.boundary_fn(self.points)

    def _refine_libstk(self):
        import ctypes

        from . import _lib
        nv, nt = len(self.points), len(self.tris)
        pts = np.ascontiguousarray(self.points, dtype=np.float64)
        tris = np.ascontiguousarray(self.tris, dtype=np.int64)
        cols = np.ascontiguousarray(self._tri_edge_color, dtype=np.int64)
        cap = 3 * nt  # untouched pages of np.empty cost nothing
        mid = np.empty((cap, 2), dtype=np.float64)
        par = np.empty((cap, 2), dtype=np.int64)
        col = np.empty(cap, dtype=np.int64)
        kids = np.empty((4 * nt, 3), dtype=np.int64)
        kcol = np.empty((4 * nt, 3), dtype=np.int64)
        ne = ctypes.c_int64()
        _lib.check(_lib.lib().stk_tri_refine(
            nv, nt, pts.ctypes.data, tris.ctypes.data, cols.ctypes.data, cap, mid.ctypes.data,
            par.ctypes.data, col.ctypes.data, kids.ctypes.data, kcol.ctypes.data, ctypes.byref(ne)))
        ne = ne.value
        self.points = np.vstack([pts, mid[:ne]])
        self.parents = np.vstack([self.parents, par[:ne]])
        self.vcolor = np.concatenate([self.vcolor, col[:ne]])
        self.nverts.append(nv + ne)
        self.tris = kids
        self._tri_edge_color = kcol
        # a pointwise test: the old vertices keep their answer
        self.boundary = np.concatenate([self.boundary, self.boundary_fn(self.points[nv:])])

    @property
    def nv(self):
        return len(self.points)

    @property
    def J(self):
        return len(self.nverts) - 1

    def levels(self):
        lv = np.zeros(self.nv, dtype=np.int64)
        for l in range(1, len(self.nverts)):
            lv[self.nverts[l - 1]:self.nverts[l]] = l
        return lv

    @property
    def cells(self):
        return self.tris


class TetMesh:
    """Red (Bey) refinements of a conforming tetrahedral mesh with hierarchical
    numbering; same attributes as TriangleMesh, with ``cells`` (nt, 4).

    The new vertices of a level are grouped by the *direction class* of the edge
    they bisect, shortest edges first (see _edge_classes for the other order), and
    sorted lexicographically inside a class.  On the Kuhn triangulation of the cube
    there are 7 classes (3 axes, 3 face diagonals, the space diagonal) and no two
    edges of a tetrahedron share one, so vertices of one class are never
    adjacent: the dof-order Gauss-Seidel sweep (reference multigrid.py:89-97) has
    at most 8 dependency levels for the 15-point mass matrix and 4 for the
    stiffness matrix, which on this mesh is the 7-point stencil (axis neighbours
    only)."""
    def __init__(self, points, tets, boundary_fn):
        self.points = np.asarray(points, dtype=np.float64)
        self.tets = np.asarray(tets, dtype=np.int64)
        self.boundary_fn = boundary_fn
        nv = len(self.points)
        self.nverts = [nv]
        self.parents = -np.ones((nv, 2), dtype=np.int64)
        self.vcolor = -np.ones(nv, dtype=np.int64)
        self.boundary = boundary_fn(self.points)

    _PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))

    def _edges(self):
        t = self.tets
        e = np.stack([t[:, list(pr)] for pr in self._PAIRS], axis=1)
        e = np.sort(e.reshape(-1, 2), axis=1)
        nv = len(self.points)
        key = e[:, 0] * nv + e[:, 1]
        ukey, inv = np.unique(key, return_inverse=True)
        edges = np.stack([ukey // nv, ukey % nv], axis=1)
        return edges, inv.reshape(-1, 6)

    def _edge_classes(self, edges):
        d = self.points[edges[:, 1]] - self.points[edges[:, 0]]
        length = np.sqrt((d * d).sum(axis=1))
        u = d / length[:, None]
        # one representative of +-u: first non-zero component positive
        u = np.round(u, 9) + 0.0
        first = np.argmax(np.abs(u) > 0, axis=1)
        sign = np.sign(u[np.arange(len(u)), first])
        u = u * sign[:, None]
        rel = np.round(length / length.min(), 9)
        # Default: shortest edges first (axes, face diagonals, space diagonal).  With
        # HYPOTENUSE_FIRST: face diagonals (two non-zero components) first, then the
        # axes, then the space diagonal -- the stiffness matrix couples axis
        # neighbours only, i.e. vertices whose numbers of odd fine-grid coordinates
        # differ by one, so {old vertices, face-diagonal midpoints} and {axis
        # midpoints, space-diagonal midpoints} are a red-black split of A_x that
        # respects the hierarchical prefix (2 dependency levels instead of 4).
        nnz = (np.abs(u) > 0).sum(axis=1)
        prio = np.where(nnz == 2, 0, np.where(nnz == 1, 1, 2)) if HYPOTENUSE_FIRST else np.zeros(len(u))
        key = np.column_stack([prio, rel, u])
        _, cls = np.unique(key, axis=0, return_inverse=True)
        return cls.reshape(-1)

    def refine(self):
        """One uniform red refinement (Bey's rule: the inner octahedron is cut
        along the diagonal x02-x13); new vertices are appended."""
        edges, te = self._edges()
        ne = len(edges)
        nv = len(self.points)
        ecls = self._edge_classes(edges)
        mid = 0.5 * (self.points[edges[:, 0]] + self.points[edges[:, 1]])
        order = np.lexsort((mid[:, 0], mid[:, 1], mid[:, 2], ecls))
        rank = np.empty(ne, dtype=np.int64)
        rank[order] = np.arange(ne)
        new_id = nv + rank

        self.points = np.vstack([self.points, mid[order]])
        self.parents = np.vstack([self.parents, edges[order]])
        self.vcolor = np.concatenate([self.vcolor, ecls[order]])
        self.nverts.append(nv + ne)

        t = self.tets
        m = new_id[te]
        x0, x1, x2, x3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        x01, x02, x03, x12, x13, x23 = (m[:, k] for k in range(6))
        self.tets = np.concatenate([
            np.stack([x0, x01, x02, x03], 1),
            np.stack([x01, x1, x12, x13], 1),
            np.stack([x02, x12, x2, x23], 1),
            np.stack([x03, x13, x23, x3], 1),
            np.stack([x01, x02, x03, x13], 1),
            np.stack([x01, x02, x12, x13], 1),
            np.stack([x02, x03, x13, x23], 1),
            np.stack([x02, x12, x13, x23], 1),
        ])
        self.boundary = self.boundary_fn(self.points)

    @property
    def cells(self):
        return self.tets

    @property
    def nv(self):
        return len(self.points)

    @property
    def J(self):
        return len(self.nverts) - 1

    def levels(self):
        lv = np.zeros(self.nv, dtype=np.int64)
        for l in range(1, len(self.nverts)):
            lv[self.nverts[l - 1]:self.nverts[l]] = l
        return lv


def _on_box_boundary(lo, hi, eps=1e-12):
    def fn(p):
        on = np.zeros(len(p), dtype=bool)
        for k in range(len(lo)):
            on |= (np.abs(p[:, k] - lo[k]) < eps) | (np.abs(p[:, k] - hi[k]) < eps)
        return on
    return fn


def construct_interval(N=16, T=1):
    """Counterpart of reference mesh.py:4-18."""
    return IntervalMesh(N, T)


def construct_2d_square_mesh(nrefines=1):
    """Unit square: 2 triangles, refined once (the analogue of the Netgen-side
    ``ngmesh.Refine()``, reference mesh.py:25-26) and then ``nrefines`` times
    (mesh.py:28-29).  The hierarchy keeps every level that has at least one
    interior vertex, so level 0 of the multigrid hierarchy has exactly 1 dof
    and the finest level has (2^(nrefines+1) - 1)^2."""
    pts = np.array([[0., 0.], [1., 0.], [1., 1.], [0., 1.]])
    tris = np.array([[0, 1, 2], [0, 2, 3]])
    mesh = TriangleMesh(pts, tris, _on_box_boundary((0., 0.), (1., 1.)))
    mesh.refine()
    for _ in range(nrefines):
        mesh.refine()
    return mesh, "default"


def construct_2d_lshape_mesh(nrefines=1):
    """L-shaped domain (-1,1)^2 \\ [0,1)x(-1,0]; not in the reference
    (reference problem.py:35-41 only knows square and cube), named by
    BASELINE.json config 4.  The coarse triangulation is deliberately not a
    three-direction mesh (alternating diagonals), so the dof order and the CSR
    rows are irregular."""
    pts = np.array([[-1., -1.], [0., -1.], [-1., 0.], [0., 0.], [1., 0.],
                    [-1., 1.], [0., 1.], [1., 1.]])
    tris = np.array([[0, 1, 3], [0, 3, 2], [2, 3, 5], [3, 6, 5], [3, 4, 7],
                     [3, 7, 6]])

    def bnd(p, eps=1e-12):
        x, y = p[:, 0], p[:, 1]
        outer = (np.abs(x + 1) < eps) | (np.abs(y - 1) < eps) | (
            (np.abs(x - 1) < eps) & (y > -eps)) | ((np.abs(y + 1) < eps) &
                                                    (x < eps))
        inner = ((np.abs(x) < eps) & (y < eps)) | ((np.abs(y) < eps) &
                                                   (x > -eps))
        return outer | inner

    mesh = TriangleMesh(pts, tris, bnd)
    mesh.refine()
    for _ in range(nrefines):
        mesh.refine()
    return mesh, "default"


def construct_3d_cube_mesh(nrefines=1):
    """Unit cube (reference mesh.py:33-43): the Kuhn triangulation (6 tetrahedra
    around the space diagonal), refined once (the analogue of the Netgen-side
    ``ngmesh.Refine()``) and then ``nrefines`` times.  Level 0 of the multigrid
    hierarchy has 1 dof, the finest level (2^(nrefines+1) - 1)^3."""
    import itertools
    pts = np.array([[x, y, z] for z in (0., 1.) for y in (0., 1.)
                    for x in (0., 1.)])
    vid = lambda v: int(v[0] + 2 * v[1] + 4 * v[2])  # noqa: E731
    tets = []
    for perm in itertools.permutations(range(3)):
        v = np.zeros(3, dtype=np.int64)
        tet = [vid(v)]
        for axis in perm:
            v[axis] = 1
            tet.append(vid(v))
        tets.append(tet)
    mesh = TetMesh(pts, np.array(tets),
                   _on_box_boundary((0., 0., 0.), (1., 1., 1.)))
    mesh.refine()
    for _ in range(nrefines):
        mesh.refine()
    return mesh, "default"
