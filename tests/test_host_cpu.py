"""Host-side logic and the C-ABI surface, without a GPU.

Covers: the built library exports every symbol include/stk.h declares; the
product path refuses host tensors (no CPU fallback); partition tables against
the reference's; wavelet split matrices against the reference's; the
Gauss-Seidel dependency schedule (its groups reproduce the sequential sweep);
the sliced-ELL builders; the duck-typed PCG / Lanczos on NumPy operands
against the oracle; the build-owned mesh / assembly identities the reference's
own tests check on NGSolve matrices."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import PKG, REPO, csr_from, load_golden, problem_from, relerr


# ---- C ABI -------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from source import _lib
    header = open(os.path.join(REPO, 'include', 'stk.h')).read()
    declared = set(re.findall(r'\b(stk_[a-z0-9_]+)\s*\(', header))
    declared -= {'stk_mg'}  # a type, not a function
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding covers the header
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(
        _lib.EXPORTED_SYMBOLS)
    assert _lib.lib().stk_version() >= 100
    assert _lib.lib().stk_last_error() is not None


def test_no_cpu_fallback():
    from source import _lib
    from source.comm import Comm
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    with pytest.raises(_lib.StkError):
        _lib.ptr(torch.zeros(4, dtype=torch.float64))
    if not torch.cuda.is_available():
        dd = DofDistributionMPI(Comm(distributed=False), 5, 3)
        x = KronVectorMPI(dd, np.ones((5, 3)))
        with pytest.raises(_lib.StkError):
            x += x  # arithmetic needs the device kernels
        with pytest.raises(_lib.StkError):
            x.dot(x)


def test_worker_threads_and_pointers_use_the_process_device(monkeypatch):
    """torch.cuda.current_device() is thread-local (a new thread starts on
    device 0): plans built in worker threads must land on the GPU pinned for the
    process, and a tensor of another GPU must be refused."""
    import threading
    from source import _lib
    monkeypatch.setattr(_lib, '_process_device', 5)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    seen = []
    worker = threading.Thread(target=lambda: seen.append(_lib.compute_device()))
    worker.start()
    worker.join()
    assert seen == [torch.device('cuda', 5)] and _lib.compute_device() == seen[0]

    class FakeTensor:
        is_cuda = True

        def __init__(self, index):
            self.device = torch.device('cuda', index)

        def data_ptr(self):
            return 4096

    assert _lib.ptr(FakeTensor(5)) == 4096
    with pytest.raises(_lib.StkError):
        _lib.ptr(FakeTensor(0))


# ---- a1: partition -------------------------------------------------------------
class _FakeComm:
    def __init__(self, rank, size):
        self.rank, self.size = rank, size

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size


def test_dof_distribution_matches_reference_tables():
    from source.mpi_vector import DofDistributionMPI
    g = load_golden('g2_partition')
    for N in (5, 9, 33, 65, 129):
        for size in (1, 2, 3, 4, 8):
            if size > N:
                continue
            tag = 'N%d_s%d' % (N, size)
            for rank in (0, size - 1):
                d = DofDistributionMPI(_FakeComm(rank, size), N, 7)
                assert np.array_equal(np.array(d.dof_distribution),
                                      g['dist_' + tag])
                assert np.array_equal(d.counts, g['counts_' + tag])
                assert np.array_equal(d.displs, g['displs_' + tag])
                assert np.array_equal(d.dof2proc, g['dof2proc_' + tag])
                assert (d.t_begin, d.t_end) == tuple(
                    g['range_%s_r%d' % (tag, rank)])
    with pytest.raises(AssertionError):
        DofDistributionMPI(_FakeComm(0, 6), 5, 7)  # more ranks than time dofs


# ---- a15: wavelet host objects ---------------------------------------------------
def test_wavelet_split_and_levels_match_reference():
    from source.wavelets import WaveletTransformOp
    g = load_golden('g1_wavelets')
    for J in range(1, 6):
        op = WaveletTransformOp(J, interleaved=True)
        assert np.array_equal(op.levels, g['levels_J%d_il' % J])
        assert op.shape == (2**J + 1, 2**J + 1)
        for j in range(J + 1):
            assert np.allclose(op.split(j).toarray(),
                               g['split_J%d_il_j%d' % (J, j)], rtol=0,
                               atol=1e-15)
        lv = WaveletTransformOp(J, interleaved=False)
        assert np.array_equal(np.asarray(lv.levels), g['levels_J%d_lv' % J])


# ---- a20: Gauss-Seidel schedule ------------------------------------------------
def _scheduled_sweep(mat, u, f, ptr, rows):
    """What the GPU does: groups in order, rows of a group from the same
    (pre-group) state."""
    invd = 1.0 / mat.diagonal()
    for g in range(len(ptr) - 1):
        idx = rows[ptr[g]:ptr[g + 1]]
        ax = mat[idx] @ u
        u[idx] += invd[idx] * (f[idx] - ax)
    return u


@pytest.mark.parametrize('fixture', ['g3_square3', 'g3_lshape', 'g3_cube'])
def test_gauss_seidel_schedule_reproduces_sequential_sweep(fixture):
    from oracle.multigrid import Smoother
    from source.multigrid import gauss_seidel_schedule
    g = load_golden(fixture)
    A = csr_from(g, 'A_x') + 0.7 * csr_from(g, 'M_x')
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    rng = np.random.RandomState(2)
    f, u0 = rng.rand(n), rng.rand(n)
    for backward in (False, True):
        ptr, rows = gauss_seidel_schedule(A.indptr, A.indices, backward)
        assert sorted(rows) == list(range(n))
        # rows of one group are mutually independent
        for k in range(len(ptr) - 1):
            grp = rows[ptr[k]:ptr[k + 1]]
            sub = A[grp][:, grp]
            assert (sub - sp.diags(sub.diagonal())).nnz == 0
        ref = u0.copy()
        sm = Smoother(A, its=1, use_c=False)
        (sm.PostSmooth if backward else sm.PreSmooth)(ref, f)
        got = _scheduled_sweep(A, u0.copy(), f, ptr, rows)
        assert relerr(got, ref) < 1e-14
    # the square numbering is built for 4 groups
    if fixture == 'g3_square3':
        assert len(gauss_seidel_schedule(A.indptr, A.indices)[0]) - 1 == 4


def test_gauss_seidel_schedule_deep_chain():
    from source.multigrid import gauss_seidel_schedule
    n = 50
    T = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1],
                 format='csr')
    ptr, rows = gauss_seidel_schedule(T.indptr, T.indices)
    assert len(ptr) - 1 == n and list(rows) == list(range(n))  # fully serial
    ptr, rows = gauss_seidel_schedule(T.indptr, T.indices, backward=True)
    assert list(rows) == list(range(n - 1, -1, -1))


# ---- ELL builders (device upload replaced by host tensors) -------------------------
@pytest.fixture
def host_uploads(monkeypatch):
    from source import _lib
    monkeypatch.setattr(_lib, 'to_dev', lambda a, dtype=None: torch.from_numpy(
        np.ascontiguousarray(a)))
    monkeypatch.setattr(_lib, 'ptr', lambda t: None if t is None else
                        t.data_ptr())


def test_union_pattern_and_ell_formats(host_uploads):
    from source.linop import (EllMatrices, EllRowsMatrix, permute_rows,
                              union_pattern)
    rng = np.random.RandomState(0)
    A = sp.random(60, 60, density=0.08, random_state=rng, format='csr')
    B = sp.random(60, 60, density=0.05, random_state=rng, format='csr')
    A = sp.csr_matrix(A + sp.identity(60))
    A[7, :30] = rng.rand(30)  # a long row -> overflow CSR in EllMatrices
    A = sp.csr_matrix(A)
    indptr, indices, (va, vb) = union_pattern([A, B])
    U = lambda v: sp.csr_matrix((v, indices, indptr), shape=A.shape)
    assert abs(U(va) - A).max() == 0 and abs(U(vb) - B).max() == 0
    order = rng.permutation(60).astype(np.int32)
    ip, ix, (pa,), rid = permute_rows(indptr, indices, [va], order)
    assert abs(sp.csr_matrix((pa, ix, ip), shape=A.shape) - A[order]).max() == 0
    assert np.array_equal(rid, order)

    A.stk_row_order = order
    ell = EllMatrices([A, B])
    assert ell.K == 16 and ell.ovf_indptr is not None
    idx = ell.ell_idx.numpy()
    for k, mat in enumerate((A, B)):
        D = np.zeros(A.shape)
        v = ell.ell_vals[k].numpy()
        for pos in range(60):
            for s in range(ell.K):
                D[order[pos], idx[pos, s]] += v[pos, s]
        op, oi, ov = (ell.ovf_indptr.numpy(), ell.ovf_indices.numpy(),
                      ell.ovf_vals[k].numpy())
        for pos in range(60):
            for e in range(op[pos], op[pos + 1]):
                D[order[pos], oi[e]] += ov[e]
        assert abs(D - mat.toarray()).max() == 0

    C = sp.csr_matrix(sp.random(40, 25, density=0.15, random_state=rng))
    C = sp.csr_matrix(C + sp.eye(40, 25))
    C.sort_indices()
    e = EllRowsMatrix(C.indptr, C.indices, C.data, None, rng.permutation(40))
    assert e.ok and e.K in EllRowsMatrix.SLOTS
    D = np.zeros(C.shape)
    ridx = e.row_ids.numpy()
    for pos in range(40):
        for s in range(e.K):
            D[ridx[pos], e.idx.numpy()[pos, s]] += e.va.numpy()[pos, s]
    assert abs(D - C.toarray()).max() == 0
    wide = sp.csr_matrix(np.ones((3, EllRowsMatrix.SLOTS[-1] + 1)))
    assert not EllRowsMatrix(wide.indptr, wide.indices, wide.data).ok


# ---- a22 / a23 on NumPy operands ---------------------------------------------------
def test_pcg_and_lanczos_duck_typed_on_numpy():
    from oracle.krylov import Lanczos as OLanczos
    from oracle.krylov import pcg as opcg
    from source.lanczos import Lanczos
    from source.linalg import PCG
    g = load_golden('g3_square')
    A = csr_from(g, 'A_x')
    Pm = sp.diags(1.0 / A.diagonal())
    b = np.random.RandomState(5).rand(A.shape[0])
    hist = []
    w, it = PCG(A, Pm, b, history=hist)
    wo, ito, ho = opcg(lambda v: A @ v, lambda v: Pm @ v, b)
    assert it == ito and np.allclose(hist, ho, rtol=1e-12)
    assert relerr(w, wo) < 1e-12
    assert relerr(A @ w, b) < 1e-5
    # early exits of reference linalg.py:17-18 and :24
    w0, it0 = PCG(A, Pm, np.zeros_like(b))
    assert it0 == 0 and not w0.any()
    lz = Lanczos(A, Pm, w=b.copy())
    lo = OLanczos(lambda v: A @ v, lambda v: Pm @ v, b.copy())
    assert lz.iterations == lo.iterations
    assert abs(lz.lmax - lo.lmax) < 1e-10 * lo.lmax
    assert abs(lz.lmin - lo.lmin) < 1e-10 * lo.lmin
    ev = np.linalg.eigvalsh((Pm @ A).toarray() if False else
                            (sp.diags(A.diagonal()**-0.5) @ A @
                             sp.diags(A.diagonal()**-0.5)).toarray())
    assert abs(lz.lmax - ev[-1]) < 1e-3 * ev[-1]
    assert abs(lz.lmin - ev[0]) < 1e-2 * ev[0]


# ---- build-owned mesh and assembly -------------------------------------------------
@pytest.mark.parametrize('name', ['square', 'lshape', 'cube'])
def test_mesh_assembly_identities(name):
    from source.assembly import (prolongation_matrices, space_load,
                                 space_matrices, tile_row_order, time_matrices)
    from source.mesh import (construct_2d_lshape_mesh,
                             construct_2d_square_mesh, construct_3d_cube_mesh,
                             construct_interval)
    from source.multigrid import gauss_seidel_schedule
    mk = {'square': construct_2d_square_mesh, 'lshape': construct_2d_lshape_mesh,
          'cube': construct_3d_cube_mesh}[name]
    J = 2 if name == 'cube' else 3
    mesh, _ = mk(J)
    M_x, A_x = space_matrices(mesh)
    n = M_x.shape[0]
    if name == 'square':
        assert n == (2**4 - 1)**2  # (2^(J+1) - 1)^2
    if name == 'cube':
        assert n == (2**3 - 1)**3
        # the Kuhn mesh: 15-point mass matrix, 7-point stiffness matrix
        assert np.diff(M_x.indptr).max() == 15 and np.diff(A_x.indptr).max() == 7
    assert M_x.dtype == np.float64 and M_x.indices.dtype == np.int32
    assert abs(M_x - M_x.T).max() < 1e-16 and abs(A_x - A_x.T).max() < 1e-13
    # mass matrix integrates: sum over free dofs of int phi_i phi_j <= area
    area = 3.0 if name == 'lshape' else 1.0
    assert 0.5 * area < M_x.sum() < area
    # stiffness of interior rows away from the boundary sums to zero
    assert np.sort(np.abs(A_x @ np.ones(n)))[n // 4] < 1e-12
    assert (np.linalg.eigvalsh(A_x.toarray())[0] > 0)
    # Galerkin products equal assembly on the coarser mesh
    # (reference multigrid_test.py:14-37)
    P = prolongation_matrices(mesh)
    assert len(P) == J and P[-1].shape[0] == n
    Mc, Ac = space_matrices(mk(J - 1)[0])
    assert abs(P[-1].T @ A_x @ P[-1] - Ac).max() < 1e-13
    assert abs(P[-1].T @ M_x @ P[-1] - Mc).max() < 1e-15
    # numbering built for shallow Gauss-Seidel dependency DAGs
    ptr, _ = gauss_seidel_schedule(M_x.indptr, M_x.indices)
    assert len(ptr) - 1 <= (8 if name == 'cube' else 5)
    order = tile_row_order(mesh)
    assert sorted(order) == list(range(n))
    # time matrices: P1 on the uniform interval
    A_t, L_t, M_t, G_t, u0_t = time_matrices(construct_interval(8))
    assert np.allclose(A_t @ np.ones(9), 0) and abs(M_t.sum() - 1.0) < 1e-15
    assert abs(L_t + L_t.T - sp.diags([-1.0] + [0.0] * 7 + [1.0])).max() < 1e-15
    assert G_t.nnz == 1 and G_t[0, 0] == 1.0 and u0_t[0] == 1.0
    tpts = np.linspace(0, 1, 9)
    assert abs(tpts @ (A_t @ tpts) - 1.0) < 1e-13  # int (t')^2 = 1
    # load vector of u0 = sin(pi x) sin(pi y) against the mass matrix
    u0 = lambda x, y: np.sin(np.pi * x) * np.sin(np.pi * y)
    if name == 'square':
        b = space_load(mesh, u0)
        pts = mesh.points[~mesh.boundary]
        assert relerr(b, M_x @ u0(pts[:, 0], pts[:, 1])) < 2e-2
    if name == 'cube':
        u3 = lambda x, y, z: u0(x, y) * np.sin(np.pi * z)
        b = space_load(mesh, u3)
        pts = mesh.points[~mesh.boundary]
        assert relerr(b, M_x @ u3(pts[:, 0], pts[:, 1], pts[:, 2])) < 5e-2


def test_p1_assembler_matches_scipy_path(monkeypatch):
    """stk_p1_assemble_2d (csrc/assemble.hip; host threads of libstk) against the
    NumPy / SciPy form of the same assembly (heateq_mpi.py:91-96,
    ngsolve_helper.py:38-45): bit for bit on the uniformly refined meshes of the
    BASELINE configurations -- the oracle fixtures were produced from the SciPy
    form --, the same pattern and values within a few ulp of the row's largest entry
    on a mesh without any symmetry, and the same matrices whatever the number of
    threads (the order of every sum is fixed by the mesh)."""
    from source.assembly import space_matrices
    from source.problem import problem_helper

    def same(a, b):
        return (a.shape == b.shape and np.array_equal(a.indptr, b.indptr)
                and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data))

    for problem, J in (('square', 1), ('square', 4), ('square', 7), ('lshape', 2), ('lshape', 6)):
        mesh = problem_helper(problem, J_space=J, J_time=2)[0]
        M1, A1 = space_matrices(mesh)
        M0, A0 = space_matrices(mesh, scipy_path=True)
        assert same(M1, M0) and same(A1, A0), (problem, J)
        assert M1.indices.dtype == np.int32 and M1.indptr.dtype == np.int32
        assert np.array_equal(M1.stk_row_order, M0.stk_row_order)
    mesh = problem_helper('lshape_jitter', J_space=6, J_time=2)[0]
    M1, A1 = space_matrices(mesh)
    M0, A0 = space_matrices(mesh, scipy_path=True)
    assert np.array_equal(A1.indices, A0.indices) and np.array_equal(M1.indices, M0.indices)
    assert np.array_equal(A1.indptr, A0.indptr)
    assert np.max(np.abs(M1.data - M0.data) / np.abs(M0.data)) < 1e-15
    assert np.max(np.abs(A1.data - A0.data)) < 1e-15 * np.abs(A0.data).max()
    assert abs(A1 - A1.T).max() < 1e-13
    for threads in ('1', '3'):
        monkeypatch.setenv('STK_HOST_THREADS', threads)
        Mt, At = space_matrices(mesh)
        assert same(Mt, M1) and same(At, A1), threads


def test_refinement_on_host_threads_matches_numpy(monkeypatch):
    """stk_tri_refine (csrc/mesh_refine.hip; host threads of libstk) against the NumPy
    form of the same refinement (source/mesh.py; stands for Netgen's Refine() and
    NGSolve's GetParentVertices, reference problem.py:7-41, multigrid.py:20-21): the
    same points, triangles, parent pairs, colours and boundary flags entry for entry,
    on the square and on the L-shape (alternating diagonals), whatever the number of
    threads -- the last levels are long enough for the sample sort's buckets.  A mesh
    whose colouring is inconsistent, or with a degenerate triangle, is refused."""
    import ctypes

    from source import _lib, mesh as M

    tables = ('points', 'tris', 'parents', 'vcolor', 'boundary', '_tri_edge_color')

    def build(make, n, numpy_path):
        orig = M.TriangleMesh.refine
        monkeypatch.setattr(M.TriangleMesh, 'refine', lambda self: orig(self, numpy_path=numpy_path))
        try:
            return make(n)[0]
        finally:
            monkeypatch.setattr(M.TriangleMesh, 'refine', orig)

    for make, n in ((M.construct_2d_square_mesh, 1), (M.construct_2d_square_mesh, 7),
                    (M.construct_2d_lshape_mesh, 6)):
        ref = build(make, n, True)
        for threads in (None, '1', '3', '7'):
            if threads is None:
                monkeypatch.delenv('STK_HOST_THREADS', raising=False)
            else:
                monkeypatch.setenv('STK_HOST_THREADS', threads)
            got = build(make, n, False)
            assert got.nverts == ref.nverts
            for name in tables:
                a, b = getattr(got, name), getattr(ref, name)
                assert a.dtype == b.dtype and np.array_equal(a, b), (make.__name__, n, threads, name)
    monkeypatch.delenv('STK_HOST_THREADS', raising=False)

    def call(pts, tris, cols):
        pts, tris, cols = (np.ascontiguousarray(pts, dtype=np.float64), np.ascontiguousarray(tris, dtype=np.int64),
                           np.ascontiguousarray(cols, dtype=np.int64))
        nt = len(tris)
        out = [np.empty((3 * nt, 2)), np.empty((3 * nt, 2), dtype=np.int64), np.empty(3 * nt, dtype=np.int64),
               np.empty((4 * nt, 3), dtype=np.int64), np.empty((4 * nt, 3), dtype=np.int64)]
        ne = ctypes.c_int64()
        return _lib.lib().stk_tri_refine(len(pts), nt, pts.ctypes.data, tris.ctypes.data, cols.ctypes.data, 3 * nt,
                                         *[o.ctypes.data for o in out], ctypes.byref(ne))

    pts = [[0., 0.], [1., 0.], [1., 1.], [0., 1.]]
    assert call(pts, [[0, 1, 2], [0, 2, 3]], [[0, 1, 2], [2, 0, 1]]) == 0
    assert call(pts, [[0, 1, 2], [0, 2, 3]], [[0, 1, 2], [2, 1, 0]]) != 0  # edge (0, 2): colour 1 and colour 0
    assert b'different colours' in _lib.lib().stk_last_error()
    assert call(pts, [[0, 1, 1], [0, 2, 3]], [[0, 1, 2], [2, 0, 1]]) != 0
    assert call(pts, [[0, 1, 4], [0, 2, 3]], [[0, 1, 2], [2, 0, 1]]) != 0


def test_load_vector_on_host_threads(monkeypatch):
    """stk_p1_load_points_2d / stk_p1_load_sum_2d (csrc/mesh_refine.hip) against the NumPy
    form of the load vector (heateq_mpi.py:102-103): the same sums in a fixed order,
    so within a few ulp of the largest entry (NumPy's matmul leaves the order to the
    BLAS); the same doubles whatever the number of host threads and however the
    evaluation of the function is sliced; a constant function gives the row sums of
    the mass matrix."""
    from source.assembly import space_load, space_matrices
    from source.problem import problem_helper

    for problem, J in (('square', 1), ('square', 6), ('lshape', 5), ('lshape_jitter', 5)):
        mesh, _, _, data, _ = problem_helper(problem, J_space=J, J_time=2)
        ref = space_load(mesh, data['u0'], numpy_path=True)
        monkeypatch.delenv('STK_HOST_THREADS', raising=False)
        got = space_load(mesh, data['u0'])
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref)) <= 1e-15 * np.max(np.abs(ref)), (problem, J)
        for threads in ('1', '5'):
            monkeypatch.setenv('STK_HOST_THREADS', threads)
            assert np.array_equal(space_load(mesh, data['u0']), got), (problem, J, threads)
        monkeypatch.delenv('STK_HOST_THREADS', raising=False)
        M_x, _ = space_matrices(mesh, scipy_path=True)
        ones = space_load(mesh, lambda x, y: 1.0)  # a scalar broadcasts over the slice
        assert ones.shape == ref.shape and np.all(ones > 0)
        # the row sums of the FULL mass matrix are |support| / 3; the free-dof block loses
        # the boundary columns, so compare where no neighbour is on the boundary
        inner = np.asarray(abs(M_x).sum(axis=1)).reshape(-1)
        full = np.isclose(inner, ones, rtol=1e-12)
        assert full.sum() > 0.5 * len(ones) or len(ones) < 16, (problem, J)


def test_host_allocator_policy():
    """source/host_malloc.py: the drivers' allocator policy (glibc: one arena, large blocks
    from the heap, no trimming) and its scoped form around a set-up.  In a child process
    -- the policy is for the life of a process: with it, blocks of 48 MB allocated and
    dropped by four threads cause next to no page faults after the first round (the heap
    is reused) where the default allocator maps and unmaps every one; STK_KEEP_MALLOC=1
    leaves the allocator alone; the scoped form nests and restores."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import os, resource, sys, threading
        sys.path.insert(0, %r)
        from source import host_malloc
        policy = host_malloc.keep_to_the_heap() if sys.argv[1] == 'heap' else False
        import numpy as np

        def faults():
            return resource.getrusage(resource.RUSAGE_SELF).ru_minflt

        def work():
            for _ in range(6):
                a = np.ones(6_000_000)
                b = a * 2
                del a, b

        def round_of_threads():
            before = faults()
            threads = [threading.Thread(target=work) for _ in range(4)]
            [t.start() for t in threads]
            [t.join() for t in threads]
            return faults() - before

        round_of_threads()
        with host_malloc.host_heap_for_setup():
            with host_malloc.host_heap_for_setup():
                inner = round_of_threads()
        print(int(policy), round_of_threads(), inner)
    """) % PKG
    out = {}
    for mode, env in (('heap', {}), ('default', {}), ('heap', {'STK_KEEP_MALLOC': '1'})):
        res = subprocess.run([sys.executable, '-c', code, mode], env=dict(os.environ, **env),
                             capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        out[(mode, bool(env))] = [int(v) for v in res.stdout.split()]
    policy, later, _ = out[('heap', False)]
    assert policy == 1
    _, later_default, _ = out[('default', False)]
    assert later * 20 < later_default, (later, later_default)  # measured: 9 against 23 000
    assert out[('heap', True)][0] == 0 and out[('heap', True)][1] * 2 > later_default


def test_plan_helpers_on_host_threads(monkeypatch):
    """stk_tile_order and stk_csr_union_count / _fill (csrc/mesh_refine.hip; host threads
    of libstk) against the NumPy / SciPy forms they stand in for on large inputs
    (source/assembly.py:tile_order_from_coords, source/linop.py:_union_pattern): the same
    arrays entry for entry, in two and three dimensions, for two and three matrices with
    explicit zeros and empty rows, whatever the number of threads; unsorted rows are
    refused."""
    import ctypes

    import scipy.sparse as sp
    from source import _lib, assembly, linop
    from source.problem import problem_helper

    rng = np.random.RandomState(3)
    # ---- tile order ----
    grid = np.stack(np.meshgrid(*[np.arange(1, 30) / 30.0] * 3, indexing='ij'), axis=-1).reshape(-1, 3)
    for problem, J in (('square', 7), ('lshape', 6), ('grid points in the cube', None)):
        if J is None:
            pts = grid[rng.permutation(len(grid))]
        else:
            mesh = problem_helper(problem, J_space=J, J_time=2)[0]
            pts = mesh.points[~mesh.boundary]
        assert len(pts) >= assembly.TILE_ORDER_ON_HOST_THREADS, (problem, len(pts))
        monkeypatch.setattr(assembly, 'TILE_ORDER_ON_HOST_THREADS', 1 << 60)
        ref = assembly.tile_order_from_coords(pts, small_lexsort=False)
        ref_small = assembly.tile_order_from_coords(pts, rows_per_tile=300, small_lexsort=False)
        monkeypatch.setattr(assembly, 'TILE_ORDER_ON_HOST_THREADS', 16384)
        for threads in (None, '1', '3'):
            if threads is None:
                monkeypatch.delenv('STK_HOST_THREADS', raising=False)
            else:
                monkeypatch.setenv('STK_HOST_THREADS', threads)
            got = assembly.tile_order_from_coords(pts, small_lexsort=False)
            assert got.dtype == ref.dtype and np.array_equal(got, ref), (problem, threads)
            assert np.array_equal(assembly.tile_order_from_coords(pts, rows_per_tile=300, small_lexsort=False),
                                  ref_small)
    monkeypatch.delenv('STK_HOST_THREADS', raising=False)
    # points that coincide in every key keep their index order (lexsort is stable)
    same = np.repeat(rng.rand(4000, 2), 5, axis=0)
    monkeypatch.setattr(assembly, 'TILE_ORDER_ON_HOST_THREADS', 1 << 60)
    ref = assembly.tile_order_from_coords(same, small_lexsort=False)
    monkeypatch.setattr(assembly, 'TILE_ORDER_ON_HOST_THREADS', 16384)
    assert np.array_equal(assembly.tile_order_from_coords(same, small_lexsort=False), ref)

    # ---- union pattern ----
    def csr32(m):
        m = sp.csr_matrix(m)
        m.sort_indices()
        m.sum_duplicates()
        m.indices, m.indptr = m.indices.astype(np.int32), m.indptr.astype(np.int32)
        return m

    n = 30000
    mats = [csr32(sp.coo_matrix((rng.rand(4 * n), (rng.randint(n, size=4 * n), rng.randint(n, size=4 * n))),
                                shape=(n, n))) for _ in range(3)]
    mats[0].data[::7] = 0.0  # explicit zeros stay entries of the pattern
    mats[1] = csr32(mats[1] + sp.identity(n, format='csr'))
    for group in (mats[:2], mats, [mats[2], mats[0]]):
        monkeypatch.setattr(linop, 'UNION_ON_HOST_THREADS', 1 << 60)
        ref = linop._union_pattern(group)
        monkeypatch.setattr(linop, 'UNION_ON_HOST_THREADS', 1)
        for threads in (None, '1', '5'):
            if threads is None:
                monkeypatch.delenv('STK_HOST_THREADS', raising=False)
            else:
                monkeypatch.setenv('STK_HOST_THREADS', threads)
            got = linop._union_pattern(group)
            assert got[0].dtype == ref[0].dtype and np.array_equal(got[0], ref[0])
            assert got[1].dtype == ref[1].dtype and np.array_equal(got[1], ref[1])
            assert len(got[2]) == len(ref[2])
            for a, b in zip(got[2], ref[2]):
                assert a.dtype == b.dtype and np.array_equal(a, b)
    monkeypatch.delenv('STK_HOST_THREADS', raising=False)
    # the assembled matrices of a mesh (what the plans unite)
    mesh = problem_helper('square', J_space=6, J_time=2)[0]
    M_x, A_x = assembly.space_matrices(mesh)
    monkeypatch.setattr(linop, 'UNION_ON_HOST_THREADS', 1 << 60)
    ref = linop._union_pattern([M_x, A_x])
    monkeypatch.setattr(linop, 'UNION_ON_HOST_THREADS', 1)
    got = linop._union_pattern([M_x, A_x])
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert all(np.array_equal(a, b) for a, b in zip(got[2], ref[2]))
    # unsorted rows
    bad = mats[0].copy()
    row = int(np.argmax(np.diff(bad.indptr) >= 2))
    lo = bad.indptr[row]
    bad.indices[lo], bad.indices[lo + 1] = bad.indices[lo + 1], bad.indices[lo]
    ptrs = (ctypes.c_void_p * 2)(bad.indptr.ctypes.data, mats[1].indptr.ctypes.data)
    idxs = (ctypes.c_void_p * 2)(bad.indices.ctypes.data, mats[1].indices.ctypes.data)
    out = np.empty(n + 1, dtype=np.int32)
    assert _lib.lib().stk_csr_union_count(n, 2, ptrs, idxs, out.ctypes.data) != 0
    assert b'ascending' in _lib.lib().stk_last_error()


def test_numbering_gives_shallow_gauss_seidel_schedules():
    """The build-owned numbering (source/mesh.py) orders the new vertices of a
    level by edge class with the hypotenuse class last: the sequential sweep of
    the reference (multigrid.py:89-97) then has 3 dependency levels for the
    5-point stiffness matrix and 4 for the 7-point mass matrix (cube: 4 and at most
    8).  With mesh.HYPOTENUSE_FIRST the stiffness matrix sweeps in 2 levels of equal
    size -- {old vertices, hypotenuse midpoints} and {short-edge midpoints} are a
    red-black split that respects the hierarchical prefix -- which is the cheaper
    but weaker smoother (measured: profiles/r04_numbering_*.log; not the default).
    Numbering inside a level is free in the reference (mesh.py:21-30)."""
    from source import mesh as mesh_mod
    from source.assembly import space_matrices
    from source.multigrid import gauss_seidel_schedule
    builds = ((mesh_mod.construct_2d_square_mesh, 3, 3, 4), (mesh_mod.construct_2d_lshape_mesh, 3, 3, 4),
              (mesh_mod.construct_3d_cube_mesh, 2, 4, 8))
    try:
        for first in (False, True):
            mesh_mod.HYPOTENUSE_FIRST = first
            for build, J, depth_a, depth_m in builds:
                M_x, A_x = space_matrices(build(J)[0], scipy_path=True)
                for backward in (False, True):
                    ptr = gauss_seidel_schedule(A_x.indptr, A_x.indices, backward)[0]
                    if first:
                        assert len(ptr) - 1 == 2 and abs(int(ptr[1]) - int(ptr[2] - ptr[1])) <= 1, ptr
                    else:
                        assert len(ptr) - 1 == depth_a, (build.__name__, ptr)
                    got = len(gauss_seidel_schedule(M_x.indptr, M_x.indices, backward)[0]) - 1
                    assert got == depth_m if depth_m == 4 else got <= depth_m
    finally:
        mesh_mod.HYPOTENUSE_FIRST = False


def test_c_partition_matches_reference_tables():
    """stk_partition (host code of libstk, no GPU needed) against the tables the
    reference's DofDistributionMPI produced (tests/golden/g2_partition.npz)."""
    import ctypes
    from source import _lib
    lib = _lib.lib()
    g = load_golden('g2_partition')
    for N in (9, 33, 65, 129):
        for size in (1, 2, 4, 8):
            tag = 'N%d_s%d' % (N, size)
            counts = (ctypes.c_int32 * size)()
            displs = (ctypes.c_int32 * size)()
            for rank in range(size):
                tb, te = ctypes.c_int32(), ctypes.c_int32()
                assert lib.stk_partition(N, size, rank, ctypes.byref(tb),
                                         ctypes.byref(te), counts, displs) == 0
                assert [tb.value, te.value] == list(g['dist_' + tag][rank])
            # the reference's counts / displs are in vector entries (rows * M)
            dist = g['dist_' + tag]
            M = g['counts_' + tag][0] / (dist[0][1] - dist[0][0])
            assert [c * M for c in counts] == list(g['counts_' + tag])
            assert [d * M for d in displs] == list(g['displs_' + tag])
    assert lib.stk_partition(3, 4, 0, None, None, None, None) != 0
    assert b'more ranks' in lib.stk_last_error()


def test_host_side_of_the_partition_independent_dot():
    """What of KronVectorMPI.dot's round-6 form runs on the host (no GPU needed): the N
    per-step sums are added in increasing t with plain additions (stk_sum_steps), the
    scratch of stk_slab_dot is a function of the slab shape alone, and the all-reduce of an
    N-vector in which every entry has ONE non-zero contributor is exact whatever the order
    of the ranks -- so the sum over the ranks followed by the sum over t is one double for
    every partition of the time axis."""
    import ctypes
    from source import _lib
    lib = _lib.lib()
    lib.stk_sum_steps.restype = ctypes.c_double
    rng = np.random.RandomState(6)
    for N in (1, 2, 9, 65, 129, 300):
        steps = np.ascontiguousarray(rng.randn(N) * 10.0**rng.randint(-8, 8, size=N))
        want = 0.0
        for v in steps:
            want += v
        assert lib.stk_sum_steps(steps.ctypes.data_as(ctypes.c_void_p), N) == want
        for size in (2, 3, 4, 8):
            if size > N:
                continue
            base, extra = divmod(N, size)
            parts, start = [], 0
            for p in range(size):
                stop = start + base + (1 if p >= size - extra else 0)
                mine = np.zeros(N)
                mine[start:stop] = steps[start:stop]
                parts.append(mine)
                start = stop
            for order in (range(size), reversed(range(size)), rng.permutation(size)):
                total = np.zeros(N)
                for p in order:
                    total = total + parts[p]
                assert np.array_equal(total, steps)
    assert lib.stk_slab_dot_work_size(1000, 65) == 2 * 33 * 4  # 33 pairs of steps x 4 blocks of 256 rows
    assert lib.stk_slab_dot_work_size(1, 1) == 2 and lib.stk_slab_dot_work_size(0, 5) == 0
    assert lib.stk_pcg_slab_work_size(1000, 65, 66, 65) == 4 * 1000 * 66 + 2 * 33 * 4 + 66


def test_thread_communicator_of_the_gpu_parity_tests():
    """tests/thread_comm.py (ranks as threads of one process: how the 8-rank shapes of the
    BASELINE configurations fit a box that allows six processes on its card) on CPU
    tensors: all-reduce, ordered point-to-point exchange with self-sends, broadcast,
    gather, barrier, and a failing rank that must not leave the others waiting."""
    import torch
    from thread_comm import run_ranks

    def body(comm):
        r, n = comm.Get_rank(), comm.Get_size()
        t = torch.zeros(n, dtype=torch.float64)
        t[r] = r + 1.0
        comm.allreduce_tensor_(t)
        assert t.tolist() == [p + 1.0 for p in range(n)]
        assert comm.allreduce(float(r)) == sum(range(n))
        # two messages to the right neighbour, one to oneself: matched in posting order
        right, left = (r + 1) % n, (r - 1) % n
        a, b, c = (torch.empty(3, dtype=torch.float64) for _ in range(3))
        reqs = comm.exchange([(torch.full((3,), 10.0 * r), right), (torch.full((3,), 10.0 * r + 1), right),
                              (torch.full((3,), -1.0 - r), r)],
                             [(a, left), (b, left), (c, r)])
        comm.wait_all(reqs)
        assert a[0] == 10.0 * left and b[0] == 10.0 * left + 1 and c[0] == -1.0 - r
        assert comm.bcast('from %d' % r, root=2) == 'from 2'
        got = comm.gather(r * r, root=1)
        assert (got == [p * p for p in range(n)]) if r == 1 else (got is None)
        comm.Barrier()
        return r

    assert run_ranks(5, body, timeout=60.0) == [0, 1, 2, 3, 4]

    def failing(comm):
        if comm.Get_rank() == 1:
            raise ValueError('rank 1 gives up')
        comm.Barrier()

    with pytest.raises(ValueError, match='rank 1 gives up'):
        run_ranks(3, failing, timeout=20.0)


# ---- property tests (hypothesis) ---------------------------------------------------
def test_properties_partition_schedule_staging():
    """Random inputs for three pieces of host logic: the slab partition, the
    Gauss-Seidel dependency schedule (groups reproduce the sequential sweep on
    arbitrary sparse matrices, forward and backward) and the zero-start entry
    filter (entries towards earlier groups only)."""
    from hypothesis import given, settings, strategies as st
    from oracle.multigrid import Smoother
    from source.comm import Comm
    from source.mpi_vector import DofDistributionMPI
    from source.multigrid import gauss_seidel_schedule

    class FakeComm(Comm):
        def __init__(self, rank, size):
            self.rank, self.size = rank, size
            self.group, self.distributed = None, False

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 300), st.integers(1, 16))
    def partition(N, size):
        size = min(size, N)
        spans = [DofDistributionMPI(FakeComm(r, size), N, 3) for r in range(size)]
        assert spans[0].t_begin == 0 and spans[-1].t_end == N
        lens = [d.t_end - d.t_begin for d in spans]
        assert all(a.t_end == b.t_begin for a, b in zip(spans, spans[1:]))
        assert max(lens) - min(lens) <= 1 and lens == sorted(lens)  # extras go last
        assert all(d.dof2proc[t] == r for r, d in enumerate(spans)
                   for t in range(d.t_begin, d.t_end))

    @settings(max_examples=40, deadline=None)
    @given(st.integers(2, 60), st.floats(0.02, 0.5), st.integers(0, 2**31 - 1),
           st.booleans())
    def schedule(n, density, seed, backward):
        rng = np.random.RandomState(seed)
        B = sp.random(n, n, density=density, random_state=rng, format='csr')
        A = sp.csr_matrix(B + B.T + sp.diags(np.abs(B + B.T).sum(axis=1).A1 + 1.0))
        A.sort_indices()
        ptr, rows = gauss_seidel_schedule(A.indptr, A.indices, backward)
        assert sorted(rows) == list(range(n))
        f, u0 = rng.rand(n), rng.rand(n)
        ref = u0.copy()
        sm = Smoother(A, its=1, use_c=False)
        (sm.PostSmooth if backward else sm.PreSmooth)(ref, f)
        assert relerr(_scheduled_sweep(A, u0.copy(), f, ptr, rows), ref) < 1e-13
        # zero start (forward): only entries towards earlier groups matter
        if not backward:
            grp = np.empty(n, dtype=int)
            for g in range(len(ptr) - 1):
                grp[rows[ptr[g]:ptr[g + 1]]] = g
            u = np.zeros(n)
            coo = A.tocoo()
            early = sp.csr_matrix((coo.data * (grp[coo.col] < grp[coo.row]),
                                   (coo.row, coo.col)), shape=A.shape)
            for g in range(len(ptr) - 1):
                idx = rows[ptr[g]:ptr[g + 1]]
                u[idx] = (f[idx] - early[idx] @ u) / A.diagonal()[idx]
            zero = np.zeros(n)
            sm.PreSmooth(zero, f)
            assert relerr(u, zero) < 1e-13

    partition()
    schedule()


@pytest.mark.parametrize('name,J', [('square', 4), ('lshape', 3), ('cube', 2)])
def test_coupling_bands_property(name, J):
    """Bands of the strip-wise smoothing: coupled rows at most one band apart,
    on every level, with coordinates (mesh rows) and without (BFS levels)."""
    from source.assembly import prolongation_matrices, space_matrices
    from source.multigrid import MeshHierarchy, coupling_bands
    from source.problem import problem_helper
    mesh = problem_helper(name, J_space=J, J_time=2)[0]
    M_x, A_x = space_matrices(mesh)
    hier = MeshHierarchy(mesh)
    mats = [sp.csr_matrix(M_x + A_x)]
    for P in reversed(hier.P_mats):
        mats.insert(0, sp.csr_matrix(P.T @ mats[0] @ P))
    for m in mats[1:]:
        n = m.shape[0]
        rows_of = np.repeat(np.arange(n), np.diff(m.indptr))
        for coords in (hier.coords, None):
            band = coupling_bands(coords, m.indptr, m.indices)
            if band is None:
                assert n < 4 or coords is None
                continue
            assert band.min() == 0 and len(band) == n
            assert np.abs(band[rows_of] - band[m.indices]).max() <= 1
            if coords is not None and n > 8:
                # as thin as the mesh rows on these structured meshes
                assert band.max() + 1 == len(np.unique(np.round(coords[:n, -1], 12)))


def test_row_grouping_of_the_packed_kronecker_form():
    """stk_pack_group_rows (host code of libstk, no device): rows that follow each
    other are grouped greedily into units of up to rp rows whose union of columns
    fits K_out slots.  Checked on random banded patterns against the definition:
    every row in exactly one unit, in order; union columns ascending and complete;
    every code where its row has the column, the zero code elsewhere; a unit ends
    only because it is full, the next row would not fit, or the rows ran out."""
    from source import _lib
    lib = _lib.lib()
    assert lib.stk_pack_unit_slots(7, 2) == 10 and lib.stk_pack_unit_slots(5, 2) == 8
    assert lib.stk_pack_unit_slots(7, 1) == 7 and lib.stk_pack_unit_slots(16, 2) == 0
    rng = np.random.RandomState(5)
    for case in range(30):
        M = int(rng.randint(1, 300))
        K = int(rng.choice([3, 5, 7]))
        rp = int(rng.randint(1, 5))
        K_out = int(rng.randint(K, 2 * K + 2))
        counts = rng.randint(0, K + 1, size=M).astype(np.int32)
        cols = np.zeros((M, K), dtype=np.int32)
        codes = np.zeros((M, K), dtype=np.int32)
        zero = 99
        rows = []
        for p in range(M):
            lo = max(0, p - 4)
            c = np.sort(rng.choice(np.arange(lo, lo + 12), size=counts[p], replace=False))
            cols[p, :counts[p]] = c
            codes[p, :counts[p]] = rng.randint(0, 50, size=counts[p])
            rows.append(dict(zip(c.tolist(), codes[p, :counts[p]].tolist())))
        own = rng.permutation(M).astype(np.int32)
        ucols = np.full((M, K_out), -7, dtype=np.int32)
        ucodes = np.full((M, K_out, rp), -7, dtype=np.int32)
        urows = np.full((M, rp), -7, dtype=np.int32)
        n_units = ctypes.c_int32()
        _lib.check(lib.stk_pack_group_rows(M, K, counts.ctypes.data, cols.ctypes.data, codes.ctypes.data,
                                           own.ctypes.data, zero, rp, K_out, ctypes.byref(n_units),
                                           ucols.ctypes.data, ucodes.ctypes.data, urows.ctypes.data))
        pos = 0
        for u in range(n_units.value):
            members = [r for r in urows[u] if r >= 0]
            n = len(members)
            assert 1 <= n <= rp and list(urows[u, n:]) == [-1] * (rp - n)
            assert members == own[pos:pos + n].tolist()  # every row once, in order
            union = sorted(set().union(*[rows[pos + j].keys() for j in range(n)]))
            assert len(union) <= K_out
            assert ucols[u, :len(union)].tolist() == union
            assert (ucols[u, len(union):] == own[pos]).all()  # unused slots: first row's own column
            for e in range(K_out):
                for j in range(rp):
                    want = rows[pos + j].get(int(ucols[u, e]), zero) if (j < n and e < len(union)) else zero
                    assert ucodes[u, e, j] == want, (case, u, e, j)
            if n < rp and pos + n < M:  # greedy: the next row did not fit
                assert len(set(union) | set(rows[pos + n].keys())) > K_out
            pos += n
        assert pos == M
    # argument errors are reported, not executed
    assert lib.stk_pack_group_rows(4, 7, None, None, None, None, 0, 2, 10, None, None, None, None) != 0
    assert b'stk_pack_group_rows' in lib.stk_last_error()


def test_newest_pmc_traffic_record_belongs_to_this_tree():
    """bench.py reports roofline.traffic from the newest record
    profiles/r*_pmc_traffic.json and refuses one measured on other kernel sources
    (round 4's driver line lost its traffic figure to a commit made two minutes after
    the PMC pass).  The record must carry the hash of the kernel file and its header as
    they stand in this tree, and the hash of the plan it streamed: a change of
    csrc/kron_pack.hip or csrc/stk_common.h without a new PMC pass
    (tools/profile_round.sh) fails HERE, on the CPU, before the round ends."""
    import glob
    import json
    import bench  # (conftest puts the repository root on sys.path)
    records = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_traffic.json')))
    assert records, 'no PMC traffic record under profiles/'
    rec = json.load(open(records[-1]))
    assert rec['source_sha'] == bench.kernel_source_sha(), (
        '%s was measured on other kernel sources: run tools/profile_round.sh on the GPU box and '
        'commit its pmc_traffic.json' % os.path.basename(records[-1]))
    assert rec.get('plan_sha'), 'the record does not name the plan it streamed'
    assert (rec['J_time'], rec['J_space'], rec['problem']) == (6, 9, 'square')  # bench.py's default workload
    assert 0.5e9 < rec['hbm_bytes_per_launch'] < 2.5e9


def test_time_factor_steps():
    """linop.time_factor_steps: the local time steps at which a tridiagonal time factor
    (three diagonals, as the kernels take them) reads its input -- what
    stk_kron_pack_apply_multi_steps is told per term.  Column t is reached through
    sub[t + 1], dia[t] and super[t - 1]; against the dense matrix's non-zero columns."""
    from source.linop import time_factor_steps
    rng = np.random.RandomState(8)
    assert time_factor_steps(None) is None
    for n in (1, 2, 5, 17):
        for trial in range(40):
            t = np.zeros((3, n))
            for _ in range(int(rng.randint(0, 3))):
                d, c = int(rng.randint(3)), int(rng.randint(n))
                if (d == 0 and c == 0) or (d == 2 and c == n - 1):
                    continue  # those entries couple to the neighbour ranks' rows, not to local steps
                t[d, c] = rng.rand() + 0.1
            T = np.diag(t[1]) + np.diag(t[0, 1:], -1) + np.diag(t[2, :-1], 1)
            cols = np.flatnonzero(np.abs(T).sum(axis=0))
            want = (0, 0) if len(cols) == 0 else (int(cols[0]), int(cols[-1]) + 1)
            assert time_factor_steps(t) == want, (n, t)
    g = np.zeros((3, 65))
    g[1, 0] = 1.0  # G_t = e_0 e_0^T (reference heateq_mpi.py:86-88)
    assert time_factor_steps(g) == (0, 1)
