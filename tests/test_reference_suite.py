"""The reference's own unit tests, one to one, run against the MI355X path.

Same test names, same inputs (the literal matrices, `arange` factors and sizes
of reference source/mpi_kron_test.py, mpi_vector_test.py, wavelets_test.py and
multigrid_test.py) and the same assertions; only the imports differ (this
package's ``source``), NGSolve's assembled matrices are replaced by the build's
own P1 assembly, and NumPy data is written through ``X_loc`` as device tensors.
The MPI variants run on one rank here; tests/test_distributed.py runs the same
operators on several ranks.
"""
import numpy as np
import pytest
import scipy.sparse
import torch

from conftest import relerr  # noqa: F401  (puts the package on sys.path)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def comm():
    from source import _lib
    _lib.lib()
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from source.comm import MPI
    return MPI.COMM_WORLD


def _fill(vec, values):
    vec.X_loc[:] = torch.from_numpy(np.ascontiguousarray(values)).to(
        vec.buf.device)


def _host(vec):
    return vec.X_loc.cpu().numpy()


# ---- mpi_kron_test.py ---------------------------------------------------------
def linearity_test_MPI(comm, linop):  # mpi_kron_test.py:12-28
    from source.mpi_kron import LinearOperatorMPI
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    assert isinstance(linop, LinearOperatorMPI)
    alpha = 3.14
    dofs_distr = DofDistributionMPI(comm, linop.N, linop.M)
    x_mpi = KronVectorMPI(dofs_distr)
    _fill(x_mpi, np.random.rand(*x_mpi.X_loc.shape))
    y_mpi = KronVectorMPI(dofs_distr)
    _fill(y_mpi, np.random.rand(*y_mpi.X_loc.shape))
    z_mpi = x_mpi + alpha * y_mpi
    result_1 = linop @ x_mpi + alpha * (linop @ y_mpi)
    result_2 = linop @ z_mpi
    assert np.allclose(_host(result_1).reshape(-1), _host(result_2).reshape(-1))


def linop_test_MPI(comm, linop_mpi, mat_glob):  # mpi_kron_test.py:31-36
    linearity_test_MPI(comm, linop_mpi)
    mat_mpi = linop_mpi.as_global_matrix()
    if comm.Get_rank() == 0:
        assert np.allclose(mat_mpi, mat_glob)


_STIFF = np.array([[3.5, 13., 28.5, 50., 77.5], [-5., -23., -53., -95., -149.],
                   [2.5, 11., 25.5, 46., 72.5]])


def _stiff_time():
    return scipy.sparse.spdiags(_STIFF, (1, 0, -1), 5, 5).T.copy().tocsr()


def test_identity_kron_mat(comm):  # mpi_kron_test.py:39-45
    from source.mpi_kron import IdentityKronMatMPI
    from source.mpi_vector import DofDistributionMPI
    N, M = 13, 16
    dofs_distr = DofDistributionMPI(comm, N, M)
    mat_space = np.arange(0, M * M).reshape(M, M)
    I_M = IdentityKronMatMPI(dofs_distr, mat_space)
    linop_test_MPI(comm, I_M, np.kron(np.eye(N), mat_space))


def test_mat_kron_identity(comm):  # mpi_kron_test.py:48-54
    from source.mpi_kron import MatKronIdentityMPI
    from source.mpi_vector import DofDistributionMPI
    N, M = 9, 16
    dofs_distr = DofDistributionMPI(comm, N, M)
    mat_time = np.arange(0, N * N).reshape(N, N)
    M_I = MatKronIdentityMPI(dofs_distr, mat_time)
    linop_test_MPI(comm, M_I, np.kron(mat_time, np.eye(M)))


def test_tridiag_kron_mat(comm):  # mpi_kron_test.py:57-66
    from source.mpi_kron import TridiagKronIdentityMPI
    from source.mpi_vector import DofDistributionMPI
    stiff_time = _stiff_time()
    M = 3
    dofs_distr = DofDistributionMPI(comm, stiff_time.shape[0], M)
    T_M = TridiagKronIdentityMPI(dofs_distr, stiff_time)
    linop_test_MPI(comm, T_M, np.kron(stiff_time.toarray(), np.eye(M)))


def test_sparse_kron_mat(comm):  # mpi_kron_test.py:69-78
    from source.mpi_kron import SparseKronIdentityMPI
    from source.mpi_vector import DofDistributionMPI
    stiff_time = _stiff_time()
    M = 3
    dofs_distr = DofDistributionMPI(comm, stiff_time.shape[0], M)
    T_M = SparseKronIdentityMPI(dofs_distr, stiff_time)
    linop_test_MPI(comm, T_M, np.kron(stiff_time.toarray(), np.eye(M)))


def test_block_diag(comm):  # mpi_kron_test.py:82-92
    from source.mpi_kron import BlockDiagMPI
    from source.mpi_vector import DofDistributionMPI
    N, M = 9, 16
    dofs_distr = DofDistributionMPI(comm, N, M)
    matrices_space = []
    np.random.seed(0)
    for n in range(N):
        matrices_space.append(np.random.rand(M, M))
    Blk = BlockDiagMPI(dofs_distr, matrices_space)
    linop_test_MPI(comm, Blk,
                   scipy.sparse.block_diag(matrices_space).toarray())


def test_composite(comm):  # mpi_kron_test.py:97-109
    from source.mpi_kron import (CompositeMPI, IdentityKronMatMPI,
                                 MatKronIdentityMPI)
    from source.mpi_vector import DofDistributionMPI
    N, M = 9, 16
    dofs_distr = DofDistributionMPI(comm, N, M)
    mat_time = np.arange(0, N * N).reshape(N, N)
    mat_space = np.arange(0, M * M).reshape(M, M)
    M_I = MatKronIdentityMPI(dofs_distr, mat_time)
    I_M = IdentityKronMatMPI(dofs_distr, mat_space)
    linop = CompositeMPI(dofs_distr, [I_M, M_I])
    composite_mat = linop.as_global_matrix()
    if comm.Get_rank() == 0:
        assert np.allclose(composite_mat, np.kron(mat_time, mat_space))


def test_sum(comm):  # mpi_kron_test.py:112-128
    from source.mpi_kron import SumMPI, TridiagKronMatMPI
    from source.mpi_vector import DofDistributionMPI
    stiff_time = _stiff_time()
    M = 3
    mat_space_1 = np.arange(0, M * M).reshape(M, M)
    mat_space_2 = np.arange(M * M, 2 * M * M).reshape(M, M)
    dofs_distr = DofDistributionMPI(comm, stiff_time.shape[0], M)
    T_M_1 = TridiagKronMatMPI(dofs_distr, stiff_time, mat_space_1)
    T_M_2 = TridiagKronMatMPI(dofs_distr, stiff_time, mat_space_2)
    T_M_sum = SumMPI(dofs_distr, [T_M_1, T_M_2])
    linop_test_MPI(comm, T_M_sum,
                   np.kron(stiff_time.toarray(), mat_space_1 + mat_space_2))


# ---- mpi_vector_test.py -------------------------------------------------------
def test_dot(comm):  # mpi_vector_test.py:7-28
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    N, M = 9, 13
    dofs_distr = DofDistributionMPI(comm, N, M)
    vec = KronVectorMPI(dofs_distr)
    x_glob = np.arange(0, N * M) * 1.0 if dofs_distr.rank == 0 else None
    vec.scatter(x_glob)
    norm_vec_sqr = vec.dot(vec)
    if dofs_distr.rank == 0:
        assert np.allclose(norm_vec_sqr, np.dot(x_glob, x_glob))
    vec_2 = KronVectorMPI(dofs_distr)
    x_glob_2 = np.random.rand(N * M) if dofs_distr.rank == 0 else None
    vec_2.scatter(x_glob_2)
    ip_vec_vec2 = vec.dot(vec_2)
    if dofs_distr.rank == 0:
        assert np.allclose(ip_vec_vec2, np.dot(x_glob, x_glob_2))


def test_permute(comm):  # mpi_vector_test.py:31-49
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    for N in range(4, 25):
        M = 244
        dofs_distr = DofDistributionMPI(comm, N, M)
        vec = KronVectorMPI(dofs_distr)
        t_glob = x_glob = None
        if dofs_distr.rank == 0:
            x_glob = np.empty(N * M, dtype=np.float64)
            t_glob = np.arange(0, N * M) * 1.0
        vec.scatter(t_glob)
        vec_space, _ = vec.permute()
        comm.Barrier()
        vec_space.gather(x_glob)
        if dofs_distr.rank == 0:
            assert np.allclose(t_glob.reshape(N, M), x_glob.reshape(M, N).T)


# ---- wavelets_test.py ---------------------------------------------------------
def test_mat_equals_matfree(comm):  # wavelets_test.py:14-19
    from source.mpi_kron import as_matrix
    from source.wavelets import WaveletTransformMat, WaveletTransformOp
    for J in range(1, 8):
        WOp = WaveletTransformOp(J)
        WMat = WaveletTransformMat(J)
        assert np.allclose(as_matrix(WOp), as_matrix(WMat))
        assert np.allclose(as_matrix(WOp.T), as_matrix(WMat).T)


def test_wavelet_transform_works(comm):  # wavelets_test.py:22-43
    from math import sqrt
    from source.wavelets import WaveletTransformOp
    J = 4
    WOp = WaveletTransformOp(J)
    I = np.eye(2**J + 1)
    # Wavelets level 0.
    assert np.allclose(WOp @ I[:, 0], np.linspace(1, 0, 2**J + 1))
    assert np.allclose(WOp @ I[:, 1], np.linspace(0, 1, 2**J + 1))
    # Wavelet of level 1.
    y = WOp @ I[:, 2]
    assert np.allclose(y[:2**(J - 1) + 1],
                       np.linspace(-sqrt(2), sqrt(2), 2**(J - 1) + 1))
    assert np.allclose(y[2**(J - 1):],
                       np.linspace(sqrt(2), -sqrt(2), 2**(J - 1) + 1))
    # Wavelet of level 2.
    y = WOp @ I[:, 3]
    assert np.allclose(y[0], -2)
    y = WOp @ I[:, 4]
    assert np.allclose(y[-1], -2)


def test_interleaved_wavelet_transform_works(comm):  # wavelets_test.py:46-72
    from math import sqrt
    from source.mpi_kron import as_matrix
    from source.wavelets import WaveletTransformOp
    WOpJ2 = WaveletTransformOp(2, interleaved=True)
    assert np.allclose(as_matrix(WOpJ2),
                       [[1, -2, -np.sqrt(2), 0, 0], [3 / 4, 2, 0, 0, 1 / 4],
                        [1 / 2, -1, np.sqrt(2), -1, 1 / 2],
                        [1 / 4, 0, 0, 2, 3 / 4], [0, 0, -np.sqrt(2), -2, 1]])
    J = 4
    WOp = WaveletTransformOp(J, interleaved=True)
    I = np.eye(2**J + 1)
    # Wavelets level 0.
    assert np.allclose(WOp @ I[:, 0], np.linspace(1, 0, 2**J + 1))
    assert np.allclose(WOp @ I[:, -1], np.linspace(0, 1, 2**J + 1))
    # Wavelet of level 1.
    y = WOp @ I[:, 2**(J - 1)]
    assert np.allclose(y[:2**(J - 1) + 1],
                       np.linspace(-sqrt(2), sqrt(2), 2**(J - 1) + 1))
    assert np.allclose(y[2**(J - 1):],
                       np.linspace(sqrt(2), -sqrt(2), 2**(J - 1) + 1))
    # Apply all the split operations after each other
    Wmat = scipy.sparse.eye(2**J + 1, 2**J + 1, format='csr')
    for j in range(1, J + 1):
        Wmat += WOp.split(j) @ Wmat
    assert np.allclose(as_matrix(WOp), as_matrix(Wmat))


def test_mpi_wavelet_transform_works(comm):  # wavelets_test.py:75-89
    from source.mpi_kron import as_matrix
    from source.mpi_vector import DofDistributionMPI
    from source.wavelets import (TransposedWaveletTransformKronIdentityMPI,
                                 WaveletTransformKronIdentityMPI,
                                 WaveletTransformOp)
    J = 4
    N = 2**J + 1
    M = 1
    dofs_distr = DofDistributionMPI(comm, N, M)
    WOp = WaveletTransformKronIdentityMPI(dofs_distr, J)
    WOpT = TransposedWaveletTransformKronIdentityMPI(dofs_distr, J)
    WOp2 = WaveletTransformOp(J, interleaved=True)
    WOpmat = WOp.as_global_matrix()
    WOpTmat = WOpT.as_global_matrix()
    if comm.Get_rank() == 0:
        assert np.allclose(WOpmat, as_matrix(WOp2))
        assert np.allclose(WOpTmat, WOpmat.T)


# ---- multigrid_test.py --------------------------------------------------------
def _stiffness(meshfn, refines):
    from source.assembly import space_matrices
    mesh, _ = meshfn(refines)
    return mesh, space_matrices(mesh)[1]


def test_prolongation(comm):  # multigrid_test.py:14-37
    from source.mesh import construct_2d_square_mesh, construct_3d_cube_mesh
    from source.mpi_kron import as_matrix
    from source.multigrid import MeshHierarchy
    for meshfn in [construct_2d_square_mesh, construct_3d_cube_mesh]:
        mesh_2, A_2 = _stiffness(meshfn, 2)
        A_mats = [_stiffness(meshfn, 0)[1], _stiffness(meshfn, 1)[1], A_2]
        hierarch = MeshHierarchy(mesh_2)
        assert hierarch.J == 2
        for j in range(hierarch.J):
            assert np.allclose(
                as_matrix(hierarch.R_mats[j] @ A_mats[j + 1] @ hierarch.P_mats[j]),
                as_matrix(A_mats[j]))


def test_smoother(comm):  # multigrid_test.py:40-58, through stk_mg_smooth
    from source import _lib
    from source.mesh import construct_2d_square_mesh
    from source.multigrid import MeshHierarchy, MultiGrid
    mesh, A = _stiffness(construct_2d_square_mesh, 2)
    mg = MultiGrid(A, MeshHierarchy(mesh))
    x = np.random.rand(A.shape[1])
    y = _lib.to_dev((A @ x).reshape(-1, 1))
    x_pre = torch.zeros_like(y)
    x_post = torch.zeros_like(y)
    for _ in range(150):
        mg.smooth(mg.hierarchy.J, x_pre, y, 1, backward=False)
        mg.smooth(mg.hierarchy.J, x_post, y, 1, backward=True)
    assert np.allclose(x_post.cpu().numpy().reshape(-1), x)
    assert np.allclose(x_pre.cpu().numpy().reshape(-1), x)


def test_multigrid_coarse(comm):  # multigrid_test.py:61-71
    from source.lanczos import Lanczos
    from source.mesh import construct_2d_square_mesh
    from source.multigrid import MeshHierarchy, MultiGrid
    mesh, A = _stiffness(construct_2d_square_mesh, 0)
    mg = MultiGrid(A, MeshHierarchy(mesh))
    lz = Lanczos(A, mg)
    assert np.allclose(lz.cond(), 1)


def test_multigrid_smoothingsteps(comm):  # multigrid_test.py:74-84
    from source.lanczos import Lanczos
    from source.mesh import construct_2d_square_mesh
    from source.multigrid import MeshHierarchy, MultiGrid
    for refines in range(3):
        mesh, A = _stiffness(construct_2d_square_mesh, refines)
        mg = MultiGrid(A, MeshHierarchy(mesh), smoothsteps=10)
        lz = Lanczos(A, mg)
        assert abs(lz.cond() - 1) < 0.01


def test_multigrid_symmetric(comm):  # multigrid_test.py:87-98
    from source.mesh import construct_2d_square_mesh
    from source.mpi_kron import as_matrix
    from source.multigrid import MeshHierarchy, MultiGrid
    for refines in range(4):
        mesh, A = _stiffness(construct_2d_square_mesh, refines)
        mg = MultiGrid(A, MeshHierarchy(mesh))
        mg_mat = as_matrix(mg)
        assert np.allclose(mg_mat.T, mg_mat)


# ---- heateq_mpi_test.py -------------------------------------------------------
refines = 2


def linop_test_apply_MPI(linop_mpi, linop):  # heateq_mpi_test.py:191-205
    from source.mpi_vector import KronVectorMPI
    np.random.seed(123123)
    x_mpi = KronVectorMPI(linop_mpi.dofs_distr)
    x_glob = y_glob = None
    if x_mpi.rank == 0:
        x_glob = np.random.rand(linop_mpi.N * linop_mpi.M)
        y_glob = linop @ x_glob
    x_mpi.scatter(x_glob)
    x_mpi = linop_mpi @ x_mpi
    x_mpi.gather(x_glob)
    if x_mpi.rank == 0:
        assert np.allclose(x_glob, y_glob)


def test_multigrid(comm):  # heateq_mpi_test.py:17-33
    from heateq_mpi import HeatEquationMPI
    from source.linop import CompositeLinOp
    from source.mpi_kron import as_matrix
    heat_eq_mpi = HeatEquationMPI(2, precond='multigrid')
    M = heat_eq_mpi.M
    for linop in [
            heat_eq_mpi.Kinv_x,
            CompositeLinOp([heat_eq_mpi.Kinv_x, heat_eq_mpi.M_x]),
            CompositeLinOp([heat_eq_mpi.Kinv_x, heat_eq_mpi.A_x])
    ]:
        x = np.random.rand(M)
        mat = as_matrix(linop)
        assert np.allclose(linop @ x, mat @ x)


def test_bilforms(comm):  # heateq_mpi_test.py:36-63
    from heateq_mpi import HeatEquationMPI
    from source.mpi_kron import as_matrix
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI

    def check_linop(N, M, linop):
        dofs_distr = DofDistributionMPI(comm, N, M)
        x_mpi = KronVectorMPI(dofs_distr)
        x_glob = y_glob = None
        if comm.Get_rank() == 0:
            x_glob = np.random.rand(N * M) * 1.0
            y_glob = np.kron(as_matrix(linop.mat_time),
                             as_matrix(linop.mat_space)) @ x_glob
        x_mpi.scatter(x_glob)
        x_mpi = linop @ x_mpi
        x_mpi.gather(x_glob)
        if comm.Get_rank() == 0:
            assert np.allclose(y_glob, x_glob)

    heat_eq_mpi = HeatEquationMPI(2)
    for linop in heat_eq_mpi.S.linops:
        check_linop(heat_eq_mpi.N, heat_eq_mpi.M, linop)


def test_matrices(comm):  # heateq_mpi_test.py:66-94
    from heateq import HeatEquation
    from heateq_mpi import HeatEquationMPI
    from source.mpi_kron import as_matrix
    J_time, J_space, problem = 4, 2, 'square'
    heat_eq_mpi = HeatEquationMPI(J_time=J_time, J_space=J_space,
                                  problem=problem, wavelettransform='original',
                                  precond='direct')
    WT_S_W_mpi = heat_eq_mpi.WT_S_W.as_global_matrix()
    P_mpi = heat_eq_mpi.P.as_global_matrix()
    if comm.Get_rank() == 0:
        heat_eq = HeatEquation(J_time=J_time, J_space=J_space, problem=problem,
                               precond='direct')
        assert np.allclose(as_matrix(heat_eq.WT_S_W), WT_S_W_mpi)
        assert np.allclose(as_matrix(heat_eq.P), P_mpi)


def test_S_apply(comm):  # heateq_mpi_test.py:97-135
    from heateq import HeatEquation
    from heateq_mpi import HeatEquationMPI
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    J_time, J_space = 4, 2
    heat_eq_mpi = HeatEquationMPI(J_time=J_time, J_space=J_space,
                                  precond='direct')
    N, M = heat_eq_mpi.N, heat_eq_mpi.M
    rank = comm.Get_rank()
    dofs_distr = DofDistributionMPI(comm, N, M)
    x_mpi = KronVectorMPI(dofs_distr)
    x_glob = y_glob = z_glob = None
    if rank == 0:
        np.random.seed(0)
        x_glob = np.random.rand(N * M) * 1.0
        heat_eq = HeatEquation(J_time=J_time, J_space=refines,
                               problem='square', precond='direct')
        y_glob = heat_eq.S @ x_glob
        S = sum([linop.as_matrix() for linop in heat_eq_mpi.S.linops])
        z_glob = S @ x_glob
    x_mpi.scatter(x_glob)
    y_mpi = heat_eq_mpi.S @ x_mpi
    y_mpi.gather(x_glob)
    if rank == 0:
        assert np.allclose(x_glob, z_glob)
        assert np.allclose(x_glob, y_glob)


def test_solve(comm):  # heateq_mpi_test.py:138-188
    import scipy.sparse
    from heateq import HeatEquation
    from heateq_mpi import HeatEquationMPI
    from source.linalg import PCG
    from source.mpi_kron import IdentityMPI
    from source.mpi_vector import DofDistributionMPI
    J_time, J_space = 4, 2
    for precond in ['direct', 'multigrid']:
        heat_eq_mpi = HeatEquationMPI(J_time=J_time, J_space=J_space,
                                      problem='square', precond=precond,
                                      smoothsteps=3, vcycles=4)
        N, M = heat_eq_mpi.N, heat_eq_mpi.M
        dofs_distr = DofDistributionMPI(comm, N, M)
        rank = comm.Get_rank()
        u_glob_mpi = f_glob_mpi = None
        if rank == 0:
            u_glob_mpi = np.empty(N * M)
            f_glob_mpi = np.empty(N * M)
            heat_eq = HeatEquation(J_time=J_time, J_space=refines,
                                   problem='square', precond='direct')
            u_glob_demo, _ = PCG(heat_eq.S, scipy.sparse.identity(N * M),
                                 heat_eq.f)
        u_mpi, _ = PCG(heat_eq_mpi.S, IdentityMPI(dofs_distr), heat_eq_mpi.rhs)
        u_mpi.gather(u_glob_mpi)
        heat_eq_mpi.rhs.gather(f_glob_mpi)
        if rank == 0:
            assert np.allclose(heat_eq.f, f_glob_mpi)
            assert np.allclose(u_glob_demo, u_glob_mpi)


def test_demo(comm):  # heateq_mpi_test.py:208-243
    from heateq import HeatEquation
    from heateq_mpi import HeatEquationMPI
    from source.mpi_kron import as_matrix
    for problem in ['square', 'cube']:
        refs = 2 if problem == 'square' else 1
        heat_eq_mpi = HeatEquationMPI(refs, precond='direct',
                                      wavelettransform='original',
                                      problem=problem)
        heat_eq = HeatEquation(problem=problem, J_space=refs, precond='direct')
        linop_test_MPI(comm, heat_eq_mpi.WT_S_W,
                       as_matrix(heat_eq.WT @ heat_eq.S @ heat_eq.W))
        linop_test_MPI(comm, heat_eq_mpi.P, as_matrix(heat_eq.P))
        for refs in range(1, 3):
            heat_eq_mpi = HeatEquationMPI(refs, precond='direct',
                                          wavelettransform='original',
                                          problem=problem)
            for op in (heat_eq_mpi.S, heat_eq_mpi.W, heat_eq_mpi.WT,
                       heat_eq_mpi.WT_S_W, heat_eq_mpi.P):
                linearity_test_MPI(comm, op)
            heat_eq = HeatEquation(problem=problem, J_space=refs,
                                   precond='direct')
            linop_test_apply_MPI(heat_eq_mpi.S, heat_eq.S)
            linop_test_apply_MPI(heat_eq_mpi.W, heat_eq.W)
            linop_test_apply_MPI(heat_eq_mpi.WT, heat_eq.WT)
            linop_test_apply_MPI(heat_eq_mpi.WT_S_W,
                                 heat_eq.WT @ heat_eq.S @ heat_eq.W)
            linop_test_apply_MPI(heat_eq_mpi.P, heat_eq.P)


def test_preconditioner(comm):  # heateq_mpi_test.py:246-274
    from heateq import HeatEquation
    from heateq_mpi import HeatEquationMPI
    from source.lanczos import Lanczos
    from source.linalg import PCG
    from source.mpi_kron import IdentityMPI
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    J_time, J_space, precond = 4, 2, 'direct'
    heat_eq_mpi = HeatEquationMPI(J_time=J_time, J_space=J_space,
                                  precond=precond)
    N, M = heat_eq_mpi.N, heat_eq_mpi.M
    dofs_distr = DofDistributionMPI(comm, N, M)
    w_mpi = KronVectorMPI(dofs_distr)
    _fill(w_mpi, np.random.rand(w_mpi.X_loc.shape[0], M))
    lanczos_mpi = Lanczos(heat_eq_mpi.WT_S_W, heat_eq_mpi.P, w=w_mpi)
    u_mpi_I, iters_I = PCG(heat_eq_mpi.S, IdentityMPI(dofs_distr),
                           heat_eq_mpi.rhs)
    u_mpi_P, iters_P = PCG(heat_eq_mpi.WT_S_W, heat_eq_mpi.P, heat_eq_mpi.rhs)
    assert iters_P < iters_I
    if w_mpi.rank == 0:
        heat_eq = HeatEquation(J_time=J_time, J_space=J_space, precond=precond)
        lanczos_demo = Lanczos(heat_eq.WT_S_W, heat_eq.P)
        assert abs(lanczos_mpi.cond() - lanczos_demo.cond()) < 0.1
