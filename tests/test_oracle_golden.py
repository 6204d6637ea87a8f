"""Pins the CPU oracle against golden vectors produced by the reference's own
classes (tests/golden/make_golden.py) and against the known-answer values in
the reference's tests.  CPU only."""
import os
from math import sqrt

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, csr_from, load_golden, problem_from, relerr
from oracle import kron, partition, wavelets
from oracle.heat import HeatEquationOracle
from oracle.krylov import Lanczos, pcg
from oracle.multigrid import MultiGrid, Smoother, galerkin_hierarchy

TOL = 1e-12


# ---- G1 + known answers: wavelets (reference wavelets_test.py) -------------
def test_wavelet_matrices_match_reference():
    g = load_golden('g1_wavelets')
    for J in range(1, 6):
        n = 2**J + 1
        I = np.eye(n)
        for inter, tag in ((True, 'il'), (False, 'lv')):
            key = 'J%d_%s' % (J, tag)
            assert np.allclose(wavelets.apply(J, I, inter), g['W_' + key],
                               rtol=0, atol=1e-14)
            assert np.allclose(wavelets.apply_transposed(J, I, inter),
                               g['WT_' + key], rtol=0, atol=1e-14)
            assert np.array_equal(wavelets.levels(J, inter),
                                  g['levels_' + key])
        assert np.allclose(wavelets.apply_stencil(J, I), g['W_J%d_il' % J],
                           rtol=0, atol=1e-14)
        assert np.allclose(wavelets.apply_transposed_stencil(J, I),
                           g['WT_J%d_il' % J], rtol=0, atol=1e-14)
        for j in range(J + 1):
            assert np.allclose(wavelets.split(J, j).toarray(),
                               g['split_J%d_il_j%d' % (J, j)], rtol=0,
                               atol=1e-15)


def test_wavelet_known_answers():
    # reference wavelets_test.py:47-51: explicit interleaved matrix for J = 2
    W2 = wavelets.apply(2, np.eye(5), interleaved=True)
    assert np.allclose(
        W2, [[1, -2, -np.sqrt(2), 0, 0], [3 / 4, 2, 0, 0, 1 / 4],
             [1 / 2, -1, np.sqrt(2), -1, 1 / 2], [1 / 4, 0, 0, 2, 3 / 4],
             [0, 0, -np.sqrt(2), -2, 1]])
    # wavelets_test.py:52-66: hat-function identities, J = 4
    J = 4
    n = 2**J + 1
    W = wavelets.apply(J, np.eye(n), interleaved=True)
    assert np.allclose(W[:, 0], np.linspace(1, 0, n))
    assert np.allclose(W[:, -1], np.linspace(0, 1, n))
    y = W[:, 2**(J - 1)]
    assert np.allclose(y[:2**(J - 1) + 1],
                       np.linspace(-sqrt(2), sqrt(2), 2**(J - 1) + 1))
    assert np.allclose(y[2**(J - 1):],
                       np.linspace(sqrt(2), -sqrt(2), 2**(J - 1) + 1))
    # wavelets_test.py:29-43: level-ordered numbering
    Wl = wavelets.apply(J, np.eye(n), interleaved=False)
    assert np.allclose(Wl[:, 0], np.linspace(1, 0, n))
    assert np.allclose(Wl[:, 1], np.linspace(0, 1, n))
    assert np.allclose(Wl[0, 3], -2) and np.allclose(Wl[-1, 4], -2)
    # wavelets_test.py:68-72: product of (I + split(j))
    Wm = sp.identity(n, format='csr')
    for j in range(1, J + 1):
        Wm = Wm + wavelets.split(J, j) @ Wm
    assert np.allclose(Wm.toarray(), W)


# ---- G2: partition ----------------------------------------------------------
def test_partition_tables_match_reference():
    g = load_golden('g2_partition')
    for N in (5, 9, 33, 65, 129):
        for size in (1, 2, 3, 4, 8):
            if size > N:
                continue
            tag = 'N%d_s%d' % (N, size)
            dist = partition.dof_distribution(N, size)
            assert np.array_equal(np.array(dist), g['dist_' + tag])
            c, d = partition.counts_displs(N, 7, size)
            assert np.array_equal(c, g['counts_' + tag])
            assert np.array_equal(d, g['displs_' + tag])
            assert np.array_equal(partition.dof2proc(N, size),
                                  g['dof2proc_' + tag])
            for rank in (0, size - 1):
                assert tuple(g['range_%s_r%d' % (tag, rank)]) == dist[rank]
    # slab sizes quoted in SURVEY.md section 8
    sizes = lambda N, s: [e - b for b, e in partition.dof_distribution(N, s)]
    assert sizes(65, 8) == [8] * 7 + [9]
    assert sizes(65, 4) == [16, 16, 16, 17]
    assert sizes(65, 2) == [32, 33]


# ---- G3: Kronecker applies -------------------------------------------------
def test_kron_applies_match_reference(g3):
    m = problem_from(g3)
    X = g3['X']
    LtT = sp.csr_matrix(m['L_t'].T)
    for nm, T, S in [('AtMx', m['A_t'], m['M_x']), ('MtAx', m['M_t'], m['A_x']),
                     ('LtAx', m['L_t'], m['A_x']), ('LtTMx', LtT, m['M_x']),
                     ('GtMx', m['G_t'], m['M_x'])]:
        assert relerr(kron.tridiag_kron_mat(T, S, X), g3['kron_' + nm]) < TOL
    metric = kron.sum_apply([(m['A_t'], m['M_x']), (m['M_t'], m['A_x'])], X)
    assert relerr(metric, g3['kron_metric']) < TOL
    # and against the dense Kronecker ground truth of mpi_kron_test.py:31-36
    dense = (kron.dense_kron(m['A_t'], m['M_x']) +
             kron.dense_kron(m['M_t'], m['A_x'])) @ X.reshape(-1)
    assert relerr(metric.reshape(-1), dense) < TOL
    assert relerr(kron.tridiag_kron_identity(m['A_t'], X),
                  g3['tridiag_At']) < TOL
    assert relerr(kron.identity_kron_mat(m['M_x'], X), g3['ident_Mx']) < TOL
    assert relerr(kron.sparse_kron_identity(m['A_t'], X), g3['sparse_At']) < TOL
    assert relerr(kron.sparse_kron_identity(m['A_t'], X, True),
                  g3['sparse_At_plusI']) < TOL
    assert relerr(kron.kron_linop(m['A_t'], m['M_x'], X.reshape(-1)),
                  g3['kronlinop_AtMx']) < TOL


def test_kron_known_answer_literal_tridiagonal():
    # reference mpi_kron_test.py:58-66, 112-128: literal spdiags data
    mat = np.array([[3.5, 13., 28.5, 50., 77.5], [-5., -23., -53., -95., -149.],
                    [2.5, 11., 25.5, 46., 72.5]])
    T = sp.spdiags(mat, (1, 0, -1), 5, 5).T.copy().tocsr()
    M = 3
    S1 = np.arange(0, M * M).reshape(M, M) * 1.0
    S2 = np.arange(M * M, 2 * M * M).reshape(M, M) * 1.0
    X = np.random.RandomState(0).rand(5, M)
    got = kron.sum_apply([(T, S1), (T, S2)], X)
    want = np.kron(T.toarray(), S1 + S2) @ X.reshape(-1)
    assert np.allclose(got.reshape(-1), want)


# ---- G4: wavelet transform kron identity -----------------------------------
def test_wavelet_kron_identity_match_reference(g3):
    J, X = int(g3['J_time']), g3['X']
    assert np.array_equal(wavelets.levels(J), g3['levels'])
    assert relerr(wavelets.apply(J, X), g3['W']) < TOL
    assert relerr(wavelets.apply_transposed(J, X), g3['WT']) < TOL
    assert relerr(wavelets.apply_stencil(J, X), g3['W']) < TOL
    assert relerr(wavelets.apply_transposed_stencil(J, X), g3['WT']) < TOL
    Y = X
    for j in range(1, J + 1):  # composite of sparse kron identity factors
        Y = kron.sparse_kron_identity(wavelets.split(J, j), Y, True)
    assert relerr(Y, g3['W']) < TOL


# ---- G8: multigrid ------------------------------------------------------------
def test_multigrid_matches_reference(g3):
    m = problem_from(g3)
    b = g3['mg_b']
    mats = galerkin_hierarchy(m['A_x'], m['P_mats'])
    for j, A in enumerate(mats):
        ref = csr_from(g3, 'galerkin_Ax_%d' % j)
        assert abs(A - ref).max() < 1e-14
    for use_c in (True, False):
        for ss in (1, 3):
            for vc in (1, 2):
                mg = MultiGrid(m['A_x'], m['P_mats'], ss, vc, use_c=use_c)
                assert relerr(mg.apply(b),
                              g3['mg_Ax_s%d_v%d' % (ss, vc)]) < TOL
    C2 = sp.csr_matrix(4 * m['M_x'] + 0.3 * m['A_x'])
    mg = MultiGrid(C2, m['P_mats'], 3, 2)
    assert relerr(mg.apply(b), g3['mg_C2_s3_v2']) < TOL
    # batch of right-hand sides == one at a time
    B = np.random.RandomState(1).rand(4, len(b))
    assert relerr(mg.apply(B)[2], mg.apply(B[2])) < 1e-14
    assert relerr((mg @ B.T)[:, 1], mg.apply(B[1])) < 1e-14


def test_smoother_converges_like_reference_test():
    # reference multigrid_test.py:40-58 (150 sweeps reach the solution)
    g = load_golden('g3_square')
    A = csr_from(g, 'A_x')
    x = np.random.RandomState(3).rand(A.shape[0])
    y = A @ x
    sm = Smoother(A, its=150)
    pre, post = np.zeros_like(x), np.zeros_like(x)
    sm.PreSmooth(pre, y)
    sm.PostSmooth(post, y)
    assert np.allclose(pre, x) and np.allclose(post, x)


# ---- G5/G6/G7: wiring, PCG, Lanczos ----------------------------------------
@pytest.mark.parametrize('precond', ['direct', 'multigrid'])
def test_heat_wiring_pcg_lanczos_match_reference(g3, precond):
    m = problem_from(g3)
    J = int(g3['J_time'])
    heat = HeatEquationOracle(m, J, precond=precond, smoothsteps=3, vcycles=2,
                              alpha=0.3)
    X = g3['X']
    assert relerr(heat.rhs(), g3['rhs']) < TOL
    assert relerr(heat.S(X), g3['S_' + precond]) < 1e-11
    assert relerr(heat.P(X), g3['P_' + precond]) < 1e-11
    assert relerr(heat.WT_S_W(X), g3['WTSW_' + precond]) < 1e-11

    # Bounds = ten times what the oracle measures against the reference's own
    # output on the four fixtures (rr <= 1.4e-12, iterate <= 4.5e-16, Lanczos
    # alpha <= 2.1e-14 over all iterations, lmax / lmin <= 2.1e-15): the GPU is
    # held to 1e-10 against ORACLE trajectories, so the oracle's own pin must
    # be tighter than that.
    rr = []
    w, iters, hist = pcg(heat.WT_S_W, heat.P, heat.rhs(),
                         callback=lambda w, r, k: rr.append(np.vdot(r, r)))
    assert iters == int(g3['pcg_iters_' + precond])
    assert np.allclose(rr, g3['pcg_rr_' + precond], rtol=2e-11, atol=0.0)
    assert relerr(w, g3['pcg_w_' + precond]) < 1e-14
    # 60 unconverged steps of plain CG on S amplify a last-bit difference of
    # one apply: 7.5e-9 on g3_square3 / multigrid, <= 1e-16 on the other seven
    w2, it2, _ = pcg(heat.S, lambda r: r.copy(), heat.rhs(), kmax=60)
    assert it2 == int(g3['pcg_unprec_iters_' + precond])
    assert relerr(w2, g3['pcg_unprec_w_' + precond]) < 1e-7

    lz = Lanczos(heat.WT_S_W, heat.P, X)
    assert lz.iterations == int(g3['lz_its_' + precond])
    n = len(g3['lz_alpha_' + precond])
    assert len(lz.alpha) == n
    assert np.allclose(lz.alpha, g3['lz_alpha_' + precond], rtol=2e-13, atol=0.0)
    assert abs(lz.lmax - g3['lz_lmax_' + precond]) < 1e-13 * lz.lmax
    assert abs(lz.lmin - g3['lz_lmin_' + precond]) < 1e-13 * lz.lmin


def test_serial_wiring_equals_parallel_wiring():
    """The serial driver's S = B^T K B + G with Y = L2_t(order 1) x H1_x
    (reference heateq.py:37-91) is the five-term Schur complement of the
    parallel driver (heateq_mpi.py:166-181): X_t and its derivative lie in Y_t.
    Pins the build's test-space time matrices and the serial oracle."""
    from oracle.heat_serial import HeatSerialOracle
    from source.assembly import (prolongation_matrices, space_load,
                                 space_matrices, time_matrices,
                                 time_matrices_test_space)
    from source.problem import problem_helper
    for problem, J_space, J_time in (('square', 2, 3), ('cube', 1, 2)):
        mesh, _, tmesh, data, _ = problem_helper(problem, J_space, J_time)
        A_t, L_t, M_t, G_t, u0_t = time_matrices(tmesh)
        M_Y, Minv_Y, B1_t, B2_t = time_matrices_test_space(tmesh)
        assert abs(M_Y @ Minv_Y - sp.eye(M_Y.shape[0])).max() < 1e-15
        assert abs(B1_t.T @ Minv_Y @ B1_t - A_t).max() < 1e-13
        assert abs(B2_t.T @ Minv_Y @ B2_t - M_t).max() < 1e-15
        assert abs(B1_t.T @ Minv_Y @ B2_t - L_t).max() < 1e-15
        M_x, A_x = space_matrices(mesh)
        mats = dict(A_t=A_t, L_t=L_t, M_t=M_t, G_t=G_t, M_x=M_x, A_x=A_x,
                    P_mats=prolongation_matrices(mesh), u0_t=u0_t,
                    u0_x=space_load(mesh, data['u0'], numpy_path=True))
        par = HeatEquationOracle(mats, J_time)
        ser = HeatSerialOracle(dict(mats, Minv_Y=Minv_Y, B1_t=B1_t, B2_t=B2_t),
                               J_time)
        X = np.random.RandomState(0).rand(par.N, par.M)
        x = X.reshape(-1)
        assert relerr(ser.S(x).reshape(X.shape), par.S(X)) < 1e-14
        assert np.array_equal(ser.f(), par.rhs().reshape(-1))
        # the serial driver numbers the wavelets level by level (heateq.py:66),
        # the parallel one interleaved (heateq_mpi.py:105-111): same operators
        # up to that permutation of the time index
        perm = np.argsort(wavelets.levels(J_time, interleaved=True),
                          kind='stable')
        assert np.array_equal(np.asarray(par.levels)[perm], ser.levels)
        Xp = X[perm]
        assert relerr(ser.P(Xp.reshape(-1)).reshape(X.shape), par.P(X)[perm]) < 1e-15
        assert relerr(ser.WT_S_W(Xp.reshape(-1)).reshape(X.shape),
                      par.WT_S_W(X)[perm]) < 1e-13


@pytest.mark.skipif(not os.path.isdir('/root/reference'),
                    reason='the reference only exists in the build container')
def test_committed_goldens_are_what_the_committed_script_produces(tmp_path):
    """Staleness guard: re-derives one fixture (g3_square, plus the wavelet and
    partition tables) by running tests/golden/make_golden.py -- i.e. the
    reference's own classes on the build's current assembly -- and compares
    every key with the committed file."""
    import subprocess
    import sys
    env = dict(os.environ, STK_GOLDEN_OUT=str(tmp_path),
               STK_GOLDEN_ONLY='g1_wavelets,g2_partition,g3_square')
    subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_golden.py')],
                   env=env, check=True, capture_output=True, timeout=900)
    for name in ('g1_wavelets', 'g2_partition', 'g3_square'):
        new = np.load(os.path.join(str(tmp_path), name + '.npz'))
        old = load_golden(name)
        assert sorted(new.files) == sorted(old.files), name
        for k in new.files:
            a, b = new[k], old[k]
            assert a.shape == b.shape, (name, k)
            if a.dtype.kind in 'iub':
                assert np.array_equal(a, b), (name, k)
            else:
                assert np.allclose(a, b, rtol=1e-12, atol=1e-300), (name, k)
