"""Worker of tests/test_distributed.py: exercises the communication layer of
KronVectorMPI over torch.distributed (gloo, host tensors).  Launched once per
rank by torch.distributed.run.  No kernel runs here (there is no CPU compute
path); what is checked is who sends which time row to whom."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

from oracle import partition, wavelets  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.mpi_vector import DofDistributionMPI, KronVectorMPI  # noqa: E402


def main():
    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    assert size == int(os.environ['WORLD_SIZE']) and size > 1
    J, M = 4, 37
    N = 2**J + 1
    dd = DofDistributionMPI(comm, N, M)
    # a1: partition equals the oracle's (pinned to the reference's tables)
    dist = partition.dof_distribution(N, size)
    assert [list(d) for d in dist] == dd.dof_distribution
    assert (dd.t_begin, dd.t_end) == dist[rank]
    assert np.array_equal(dd.dof2proc, partition.dof2proc(N, size))
    c, d = partition.counts_displs(N, M, size)
    assert np.array_equal(dd.counts, c) and np.array_equal(dd.displs, d)

    X = np.random.RandomState(7).rand(N, M)  # same on every rank
    v = KronVectorMPI(dd)
    assert not v.buf.is_cuda or os.environ.get('STK_BACKEND') == 'gloo'
    # a6: scatter / gather
    v.scatter(X.reshape(-1) if rank == 0 else None)
    assert np.array_equal(v.X_loc.cpu().numpy(), X[dd.t_begin:dd.t_end])
    out = np.zeros(N * M) if rank == 0 else None
    v.gather(out)
    if rank == 0:
        assert np.array_equal(out.reshape(N, M), X)
    # a4: halo rows
    t = v.communicate_bdr()
    assert t >= 0.0 and v.communicated_bdr
    if rank > 0:
        assert np.array_equal(v.X_lo.cpu().numpy(), X[dd.t_begin - 1])
    else:
        assert v.X_lo is None
    if rank + 1 < size:
        assert np.array_equal(v.X_hi.cpu().numpy(), X[dd.t_end])
    else:
        assert v.X_hi is None
    called = []
    assert v.communicate_bdr(lambda: called.append(1)) == 0.0 and called  # cached
    v._invalidate()
    assert not v.communicated_bdr
    # the same rows cut into pieces that travel through intermediate ranks (two
    # hops; opt-in STK_HALO_ROUTES): every route count up to the ranks there are
    from source.mpi_vector import halo_route_plan
    for routes in (2, 3, 7):
        KronVectorMPI.HALO_ROUTES = routes
        try:
            v._invalidate()
            if v._ghost is not None:
                v._ghost.fill_(-1.0)
            called = []
            v.communicate_bdr(lambda: called.append(1))
            assert called == [1]
            if rank > 0:
                assert np.array_equal(v.X_lo.cpu().numpy(), X[dd.t_begin - 1]), routes
            if rank + 1 < size:
                assert np.array_equal(v.X_hi.cpu().numpy(), X[dd.t_end]), routes
        finally:
            KronVectorMPI.HALO_ROUTES = 1
        plan = halo_route_plan(size, M, routes)
        if size > 2:  # pieces really leave the direct link
            assert any(ph == 2 for ph, *_ in plan), (size, routes)
        # every row arrives whole: the pieces delivered to a rank tile [0, M)
        for dest in range(size):
            for direction in (+1, -1):
                owner = dest - direction
                if 0 <= owner < size:
                    got = sorted((a, b) for ph, s_, d_, o_, dr, a, b in plan
                                 if d_ == dest and o_ == owner and dr == direction
                                 and d_ == o_ + dr)
                    assert got[0][0] == 0 and got[-1][1] == M
                    assert all(x[1] == y[0] for x, y in zip(got, got[1:]))
    v._invalidate()
    # the start-up probe that chooses the halo form on real rows: collective, every
    # rank ends with the same choice, a routed form is only ever taken when it
    # delivered the direct exchange's rows bit for bit and was faster
    from source.mpi_vector import probe_halo_form, startup_report
    rec = probe_halo_form(dd)
    assert rec['chosen'] == KronVectorMPI.HALO_ROUTES
    assert comm.allreduce(float(rec['chosen'])) == size * rec['chosen']
    if size <= 2:
        assert rec['chosen'] == 1 and not rec['identical']
    else:
        assert set(rec['identical']) == {k for k in (3, 7) if k <= size - 1}
        assert all(rec['identical'].values()), rec  # gloo delivers the routed rows intact
        assert rec['chosen'] == 1 or rec['ms'][rec['chosen']] < rec['ms'][1]
    KronVectorMPI.HALO_ROUTES = 1
    # the policy around the probe (choose_halo_form): the variable unset means the probe
    # on gloo (and the direct exchange on RCCL, where the routed form has never run); a
    # pinned value is taken as it is, without a probe
    from source.mpi_vector import choose_halo_form
    assert 'STK_HALO_ROUTES' not in os.environ
    pol = choose_halo_form(dd)
    assert 'probe (gloo)' in pol['policy'] and pol['chosen'] == KronVectorMPI.HALO_ROUTES
    os.environ['STK_HALO_ROUTES'] = '1'
    try:
        pol = choose_halo_form(dd)
    finally:
        del os.environ['STK_HALO_ROUTES']
    assert pol['chosen'] == 1 and pol['reason'] == 'not probed' and KronVectorMPI.HALO_ROUTES == 1
    # the counters of the scalar all-reduce (what the drivers report per iteration)
    type(comm).timing = True
    comm.reset_counters()
    for _ in range(3):
        comm.allreduce_tensor_(torch.ones(1, dtype=torch.float64))
    type(comm).timing = False
    assert comm.allreduce_calls == 3 and comm.allreduce_host_s > 0.0
    info = startup_report(dd, [v.buf])
    assert info['rank'] == rank and info['backend'] == 'gloo'
    v._invalidate()
    # a5: arbitrary remote rows, pattern of every wavelet level
    for j in range(1, J + 1):
        S = wavelets.split(J, j).tocoo()
        pairs = sorted(set((int(r), int(c)) for r, c in zip(S.row, S.col)
                           if dd.t_begin <= r < dd.t_end and not
                           (dd.t_begin <= c < dd.t_end)))
        recv, slot, reqs = v.communicate_dofs(pairs)
        comm.wait_all(reqs)
        for row, k in slot.items():
            assert np.array_equal(recv[k].cpu().numpy(), X[row]), (j, row)
    # a3: scalar all-reduce
    assert comm.allreduce(float(rank + 1)) == size * (size + 1) / 2
    # permute: all-to-all transpose and back (reference mpi_vector_test.py:31-48)
    vp, _ = v.permute()
    assert vp.N == M and vp.M == N
    assert np.array_equal(vp.X_loc.cpu().numpy(), X.T[vp.t_begin:vp.t_end])
    back, _ = vp.permute()
    assert np.array_equal(back.X_loc.cpu().numpy(), X[dd.t_begin:dd.t_end])
    comm.Barrier()
    if rank == 0:
        print('mp_comm_worker ok, size', size)


if __name__ == '__main__':
    main()
