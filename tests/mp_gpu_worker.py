"""Worker of tests/test_distributed.py (gpu): the time-slab distributed solver
on real kernels.  Several ranks share the one GPU of the test box and talk
through gloo (STK_BACKEND=gloo, device buffers staged through the host); the
data path -- ghost lanes of the Kronecker kernel, per-level wavelet exchange,
scalar all-reduce -- is the one RCCL carries on an 8-GPU node.  Rank 0 checks
everything against the CPU oracle."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

import heateq_mpi as hm  # noqa: E402
from oracle import kron as okron  # noqa: E402
from oracle.heat import HeatEquationOracle  # noqa: E402
from oracle.krylov import pcg  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.linalg import PCG  # noqa: E402
from source.mpi_kron import SumMPI, TridiagKronMatMPI  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402


def main():
    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    J_time, J_space = int(os.environ.get('STK_TEST_J_TIME', '4')), 3
    problem = os.environ.get('STK_TEST_PROBLEM', 'square')
    h = hm.HeatEquationMPI(J_space=J_space, J_time=J_time, problem=problem)
    dd = h.dofs_distr
    N, M = h.N, h.M
    X = np.random.RandomState(128).rand(N, M)

    def scattered():
        v = KronVectorMPI(dd)
        v.scatter(X.reshape(-1) if rank == 0 else None)
        return v

    def gathered(v):
        out = np.zeros(N * M) if rank == 0 else None
        v.gather(out)
        return out.reshape(N, M) if rank == 0 else None

    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    o = None
    if rank == 0:
        mats = dict(A_t=h.A_t, L_t=h.L_t, M_t=h.M_t, G_t=h.G_t, M_x=h.M_x,
                    A_x=h.A_x, P_mats=h.hierarchy.P_mats, u0_t=h.u0_t,
                    u0_x=h.u0_x)
        o = HeatEquationOracle(mats, J_time)

    x = scattered()
    metric = SumMPI(dd, [TridiagKronMatMPI(dd, h.A_t, h.M_x),
                         TridiagKronMatMPI(dd, h.M_t, h.A_x)])
    got = gathered(metric @ x)
    if rank == 0:
        want_metric = okron.sum_apply([(h.A_t, h.M_x), (h.M_t, h.A_x)], X)
        assert rel(got, want_metric) < 1e-12, ('metric', rel(got, want_metric))
    for name, op, ref in [('W', h.W, 'W'), ('WT', h.WT, 'WT'), ('S', h.S, 'S'),
                          ('P', h.P, 'P'), ('WTSW', h.WT_S_W, 'WT_S_W')]:
        x._invalidate()
        got = gathered(op @ x)
        if rank == 0:
            want = getattr(o, ref)(X)
            assert rel(got, want) < 1e-11, (name, rel(got, want))
    # the reference's per-level wavelet composite agrees with the transposes
    for op, ref in ((h.W, 'W'), (h.WT, 'WT')):
        op.mode = 'composite'
        x._invalidate()
        got = gathered(op @ x)
        op.mode = 'transpose'
        if rank == 0:
            assert rel(got, getattr(o, ref)(X)) < 1e-11, ('composite', ref)
    # the reference-structured S (5 terms, time factor then space operator)
    h2 = hm.HeatEquationMPI(J_space=J_space, J_time=J_time, schur='reference',
                            problem=problem)
    got = gathered(h2.S @ scattered())
    if rank == 0:
        assert rel(got, o.S(X)) < 1e-11
    # dot + solve
    d = x.dot(x)
    assert abs(d - np.vdot(X, X)) < 1e-12 * d
    hist = []
    w, its = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
    got = gathered(w)
    if rank == 0:
        wo, it_o, hist_o = pcg(o.WT_S_W, o.P, o.rhs())
        assert its == it_o, (its, it_o)
        assert np.allclose(hist, hist_o, rtol=1e-10, atol=1e-30), float(np.max(np.abs(np.asarray(hist) / np.asarray(hist_o) - 1)))
        assert rel(got, wo) < 1e-8
    # the reference-arithmetic mode across ranks (general block-diagonal path: one
    # hierarchy per wavelet level, time slices gathered per operator and rank)
    h3 = hm.HeatEquationMPI(J_space=J_space, J_time=J_time, problem=problem, arithmetic='reference')
    hist3 = []
    w3, its3 = PCG(h3.WT_S_W, h3.P, h3.rhs, history=hist3)
    got3 = gathered(w3)
    if rank == 0:
        assert its3 == it_o, (its3, it_o)
        assert np.allclose(hist3, hist_o, rtol=1e-10, atol=1e-30)
        assert rel(got3, wo) < 1e-9
    # both forms of the halo on the packed path give the same operator
    # (the overlapped form is the default from 24 steps on: taken here on every slab)
    from source.mpi_kron import _FusedKronSum
    default_from = _FusedKronSum.OVERLAP_FROM
    _FusedKronSum.OVERLAP_FROM = 1
    try:
        x._invalidate()
        y_overlap = gathered(metric @ x)
        _FusedKronSum.overlap = False
        x._invalidate()
        y_one_pass = gathered(metric @ x)
    finally:
        _FusedKronSum.overlap = True
        _FusedKronSum.OVERLAP_FROM = default_from
    if rank == 0:
        assert np.array_equal(y_overlap, y_one_pass) and rel(y_overlap, want_metric) < 1e-12
    # the mirrored driver end to end on the same ranks (reference heateq_mpi.py:205-312):
    # its first-contact record -- start-up line per rank, on 3 ranks and more the probe
    # that chooses the halo form -- and the same solve
    if os.environ.get('STK_HALO_ROUTES') is None:
        _, sol_d, its_d, hist_d = hm.main(['--J_time=%d' % J_time, '--J_space=%d' % J_space,
                                          '--problem=%s' % problem])
        assert its_d == its and np.allclose(hist_d, hist, rtol=1e-12, atol=0.0)
        KronVectorMPI.HALO_ROUTES = 1
    if rank == 0:
        print('mp_gpu_worker ok: size %d, %d PCG iterations' % (size, its))
    comm.Barrier()


if __name__ == '__main__':
    main()
