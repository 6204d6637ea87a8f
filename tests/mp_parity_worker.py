"""Worker of tests/test_distributed.py (gpu): a BASELINE configuration solved on
several ranks, compared BIT FOR BIT with the one-rank solve of the same process and
within the north star's 1e-10 with the CPU oracle's trajectory
(tests/golden/o1_pcg_*; reference heateq_mpi_test.py:138-189 runs its comparison
under mpirun the same way).

Two ways to be several ranks on the one GPU of the test box:
* launched by torch.distributed.run with STK_BACKEND=gloo: one process per rank
  (the pool allows six processes on a card: up to five ranks);
* STK_TEST_THREAD_RANKS=R: R ranks as threads of this process
  (tests/thread_comm.py) -- how the 8-rank shapes of configs 2 and 3 fit the box.

What must hold on any number of ranks, because every kernel rounds every entry as
the one-rank kernel does and dot sums per time step in a fixed shape:
iteration count, the whole r.Pr history, the iterate and the metric operator's
output are EQUAL to the one-rank run's (np.array_equal)."""
import os
import sys
import threading

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

import heateq_mpi as hm  # noqa: E402
from source.comm import MPI, Comm  # noqa: E402
from source.linalg import PCG  # noqa: E402
from source.mpi_kron import SumMPI, TridiagKronMatMPI  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402

J_TIME = int(os.environ.get('STK_TEST_J_TIME', '5'))
J_SPACE = int(os.environ.get('STK_TEST_J_SPACE', '8'))
PROBLEM = os.environ.get('STK_TEST_PROBLEM', 'square')
# 'composite' (the default; the oracle's trajectories were produced with it), 'original' or
# 'interleaved': the wavelet transform as a matrix between two all-to-all exchanges
WAVELETS = os.environ.get('STK_TEST_WAVELETS', 'composite')
LANCZOS = os.environ.get('STK_TEST_LANCZOS') == '1'
PRECOND = os.environ.get('STK_TEST_PRECOND', 'multigrid')  # or 'direct' (heateq_mpi.py:155-157)
SETUP = threading.Lock()  # plan construction reads process-wide tuning keys: one rank at a time
# The overlapped halo form (pass without the ghost steps beside the exchange, boundary
# steps recomputed afterwards from the records the pack leaves) is the default; one case of
# the test asks for the other form (STK_TEST_OVERLAP_FROM above every slab length: wait, one
# pass with ghost lanes): both are bit for bit the one-rank kernel.
from source.mpi_kron import _FusedKronSum  # noqa: E402
_FusedKronSum.OVERLAP_FROM = int(os.environ.get('STK_TEST_OVERLAP_FROM', '1'))
_X = {}


def bench_vector(N, M):
    """The bench's input, np.random.seed(128); rand(N, M) (heateq_mpi_timing.py:81-83)."""
    with SETUP:
        if (N, M) not in _X:
            _X[(N, M)] = np.random.RandomState(128).rand(N, M)
    return _X[(N, M)]


def solve(comm):
    """(iterations, history, iterate, metric output) -- the last two gathered on rank 0."""
    with SETUP:
        h = hm.HeatEquationMPI(J_space=J_SPACE, J_time=J_TIME, problem=PROBLEM, comm=comm,
                               wavelettransform=WAVELETS, precond=PRECOND)
    dd = h.dofs_distr
    X = bench_vector(h.N, h.M)
    x = KronVectorMPI(dd, X[dd.t_begin:dd.t_end])
    metric = SumMPI(dd, [TridiagKronMatMPI(dd, h.A_t, h.M_x), TridiagKronMatMPI(dd, h.M_t, h.A_x)])

    def gathered(v):
        out = np.zeros(h.N * h.M) if comm.Get_rank() == 0 else None
        v.gather(out)
        return out

    y = gathered(metric @ x)
    del metric
    hist = []
    w, its = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
    out = (its, np.asarray(hist), gathered(w), y, (h.N, h.M))
    if LANCZOS:
        # the condition-number estimate of the preconditioned system (reference
        # lanczos.py:9-171) from the bench vector: its recurrence is inner products and
        # applies only, so its coefficients do not see the partition either
        from source.lanczos import Lanczos
        lz = Lanczos(h.WT_S_W, h.P, w=x)
        k = int(np.flatnonzero(lz.alpha)[-1]) + 1
        out = out + ((lz.alpha[:k].copy(), lz.beta[:k - 1].copy(), lz.lmax, lz.lmin),)
    return out


def main():
    threads = int(os.environ.get('STK_TEST_THREAD_RANKS', '0'))
    if threads:
        from thread_comm import run_ranks
        size = threads
        got = run_ranks(size, solve)[0]
        rank = 0
    else:
        comm = MPI.COMM_WORLD
        rank, size = comm.Get_rank(), comm.Get_size()
        assert size > 1
        got = solve(comm)
        comm.Barrier()
    if rank != 0:
        return
    torch.cuda.empty_cache()
    its, hist, w, y, (N, M) = got[:5]
    one = solve(Comm(distributed=False))
    its1, hist1, w1, y1, _ = one[:5]
    if LANCZOS:
        (alpha, beta, lmax, lmin), (alpha1, beta1, lmax1, lmin1) = got[5], one[5]
        assert len(alpha) == len(alpha1) > 3, (len(alpha), len(alpha1))
        assert np.array_equal(alpha, alpha1) and np.array_equal(beta, beta1), 'Lanczos coefficients'
        assert lmax == lmax1 and lmin == lmin1 and 0 < lmin < lmax, (lmax, lmin)
        print('Lanczos: %d steps, lmax %.6f lmin %.6f, equal to the one-rank run' % (len(alpha), lmax, lmin))
    # the partition of the time axis leaves no trace
    assert np.array_equal(y, y1), ('metric operator', float(np.max(np.abs(y - y1))))
    assert its == its1, (its, its1)
    assert np.array_equal(hist, hist1), ('history', hist / hist1 - 1.0)
    assert np.array_equal(w, w1), ('iterate', float(np.max(np.abs(w - w1))))
    if WAVELETS != 'composite':  # another basis ordering: no trajectory of the oracle to compare with
        print('mp_parity_worker ok: %s J_time=%d J_space=%d on %d %s, %s wavelets, %d iterations, history '
              'equal to the one-rank run' % (PROBLEM, J_TIME, J_SPACE, size,
                                             'threads' if threads else 'processes', WAVELETS, its))
        return
    # ... and the trajectory is the CPU path's
    g = np.load(os.path.join(HERE, 'golden', 'o1_pcg_%s_J%d_J%d%s.npz' % (
        PROBLEM, J_TIME, J_SPACE, '' if PRECOND == 'multigrid' else '_' + PRECOND)))
    assert its == int(g['iters']), (its, int(g['iters']))
    dev = float(np.max(np.abs(hist / g['hist'] - 1.0)))
    assert dev < 1e-10, dev
    st, sx = (int(v) for v in g['sample_strides'])
    w = w.reshape(N, M)
    assert abs(np.linalg.norm(w) - g['w_norm']) < 1e-10 * g['w_norm']
    err = np.linalg.norm(w[::st, ::sx] - g['w_sample']) / np.linalg.norm(g['w_sample'])
    assert err < 1e-10, err
    out = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out):  # the record DESIGN.md section 5 quotes
        import json
        path = os.path.join(out, 'parity_multi_rank.json')
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec['%s_J%d_J%d_%d_%s%s' % (PROBLEM, J_TIME, J_SPACE, size, 'threads' if threads else 'processes',
                                     '' if PRECOND == 'multigrid' else '_' + PRECOND)] = {
            'iterations': int(its), 'history_equal_to_one_rank': True, 'iterate_equal_to_one_rank': True,
            'metric_operator_equal_to_one_rank': True, 'max_rel_dev_from_oracle_history': dev,
            'iterate_sample_rel_err_vs_oracle': float(err)}
        json.dump(rec, open(path, 'w'), indent=1, sort_keys=True)
    print('mp_parity_worker ok: %s J_time=%d J_space=%d on %d %s, %d iterations, history equal to the '
          'one-rank run, %.2e from the oracle' % (PROBLEM, J_TIME, J_SPACE, size,
                                                  'threads' if threads else 'processes', its, dev))


if __name__ == '__main__':
    main()
