"""Multi-process tests of the time-slab distribution.

* CPU, gloo, world size 2 and 3: the communication layer (partition, halo,
  strided partner rows, all-to-all transpose, scalar all-reduce).
* GPU: 2 and 3 ranks sharing the box's one GPU run the real kernels and the
  whole preconditioned solve; rank 0 compares with the CPU oracle.
"""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(worker, nproc, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()),
           os.path.join(HERE, worker)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return res.stdout


@pytest.mark.parametrize('nproc', [2, 3, 5, 8])
def test_comm_layer_gloo_cpu(nproc):
    out = _run('mp_comm_worker.py', nproc, {'HIP_VISIBLE_DEVICES': '',
                                            'CUDA_VISIBLE_DEVICES': ''})
    assert 'mp_comm_worker ok' in out


@pytest.mark.gpu
@pytest.mark.parametrize('nproc,problem', [(2, 'square'), (3, 'square'),
                                           (4, 'lshape'), (5, 'square')])
def test_distributed_solve_ranks_sharing_one_gpu(nproc, problem):
    out = _run('mp_gpu_worker.py', nproc, {'STK_BACKEND': 'gloo',
                                           'STK_TEST_PROBLEM': problem})
    assert 'mp_gpu_worker ok' in out


@pytest.mark.gpu
def test_distributed_solve_with_routed_halo():
    """The whole multi-rank worker (operators, both halo forms, solve in both
    arithmetic modes) with every halo row cut into three pieces that travel through
    intermediate ranks (opt-in STK_HALO_ROUTES)."""
    out = _run('mp_gpu_worker.py', 4, {'STK_BACKEND': 'gloo', 'STK_HALO_ROUTES': '3'})
    assert 'mp_gpu_worker ok' in out


@pytest.mark.gpu
def test_distributed_solve_on_slabs_long_enough_for_row_pairs():
    """Two ranks with 32 and 33 time steps each (J_time = 6): the Kronecker
    operators then run the row-pair form of the packed kernel together with its
    ghost lanes (shorter slabs keep one row per slot row)."""
    out = _run('mp_gpu_worker.py', 2, {'STK_BACKEND': 'gloo', 'STK_TEST_J_TIME': '6'})
    assert 'mp_gpu_worker ok' in out


@pytest.mark.gpu
@pytest.mark.parametrize('nproc', [2, 4])
def test_baseline_config_on_several_processes_equals_the_one_rank_solve(nproc):
    """BASELINE config 2 (J_time = 5, J_space = 8, square) on 2 and 4 processes sharing
    the GPU over gloo: iteration count, the whole r.Pr history, the iterate and the
    metric operator's output are EQUAL (bit for bit) to the one-rank run, and the
    history is within 1e-10 of the CPU oracle's (tests/golden/o1_pcg_square_J5_J8):
    reference heateq_mpi_test.py:138-189 under mpirun."""
    out = _run('mp_parity_worker.py', nproc, {'STK_BACKEND': 'gloo', 'STK_TEST_J_TIME': '5',
                                              'STK_TEST_J_SPACE': '8'}, timeout=1200)
    assert 'mp_parity_worker ok' in out


@pytest.mark.gpu
@pytest.mark.parametrize('J_time,J_space,ranks', [(5, 8, 8), (6, 9, 8), (6, 9, 3), (7, 10, 8)])
def test_baseline_config_in_its_eight_rank_shape_equals_the_one_rank_solve(J_time, J_space, ranks):
    """Configs 2 and 3 cut into EIGHT time slabs (config 3 is defined as an 8-GPU
    run: 9-step slabs of 1 046 529 rows; config 2: slabs of 4 and 5 steps), every rank a
    thread of one process on the box's GPU (tests/thread_comm.py -- the pool allows
    six processes on a card): halo exchange, the overlapped ghost form, the all-to-all
    transposes of the wavelet transform and the per-step partial sums of dot as on an
    8-GPU node.  The solve is bit for bit the one-rank solve and within 1e-10 of the
    oracle's trajectory; config 3 also on three ranks (slabs of 21, 22, 22), and CONFIG 5
    (J_time = 7, J_space = 10) in the shape BASELINE.json names it in: eight slabs of 16 and
    17 steps of 4 190 209 rows (72 s on the box; eight sets of plans on the one card)."""
    env = dict(os.environ, STK_TEST_THREAD_RANKS=str(ranks), STK_TEST_J_TIME=str(J_time),
               STK_TEST_J_SPACE=str(J_space), OMP_NUM_THREADS='1')
    if ranks == 3:  # the OTHER halo form: wait for the rows, one pass with ghost lanes
        env['STK_TEST_OVERLAP_FROM'] = '1000'
    res = subprocess.run([sys.executable, os.path.join(HERE, 'mp_parity_worker.py')], env=env,
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert 'mp_parity_worker ok' in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize('problem,J_time,J_space,ranks,wavelets', [
    ('lshape', 5, 8, 8, 'composite'), ('square', 5, 8, 3, 'original'), ('square', 5, 8, 5, 'interleaved'),
    ('square', 3, 6, 4, 'direct')])
def test_other_shapes_of_the_solve_on_several_ranks_equal_the_one_rank_solve(problem, J_time, J_space, ranks, wavelets):
    """What the eight-rank tests above leave out: BASELINE config 4 (the L-shape, J_time = 5,
    J_space = 8: matrices without the square's repeated values, i.e. the plans with explicit
    values) cut into eight slabs, and the wavelet transform as a matrix between two
    all-to-all exchanges (heateq_mpi.py:126-139 'original' and 'interleaved':
    MatKronIdentityMPI, mpi_kron.py:225-256) on three and five ranks -- each bit for bit
    the one-rank solve; the L-shape also within 1e-10 of the oracle's trajectory, and with
    the coefficients and the extreme Ritz values of the Lanczos recurrence on the
    preconditioned system equal to the one-rank run's.  Last case: config 1 with
    precond='direct' (heateq_mpi.py:155-157; InvLinOp's factors on the device)
    on four ranks, against tests/golden/o1_pcg_square_J3_J6_direct."""
    env = dict(os.environ, STK_TEST_THREAD_RANKS=str(ranks), STK_TEST_J_TIME=str(J_time),
               STK_TEST_J_SPACE=str(J_space), STK_TEST_PROBLEM=problem, STK_TEST_WAVELETS=wavelets,
               OMP_NUM_THREADS='1')
    if problem == 'lshape':  # ... and the Lanczos recurrence (lanczos.py:9-171) on the eight slabs
        env['STK_TEST_LANCZOS'] = '1'
    if wavelets == 'direct':  # config 1 with the direct preconditioner: SuperLU's factors on every rank's device
        env.update(STK_TEST_WAVELETS='composite', STK_TEST_PRECOND='direct')
    res = subprocess.run([sys.executable, os.path.join(HERE, 'mp_parity_worker.py')], env=env,
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert 'mp_parity_worker ok' in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize('nproc', [1, 2, 3])
def test_every_operator_class_on_several_ranks(nproc):
    """Every operator class and the vector algebra against dense NumPy ground
    truth on 1, 2 and 3 ranks sharing the GPU (the reference runs its unit
    tests under mpirun the same way)."""
    out = _run('mp_ops_worker.py', nproc, {'STK_BACKEND': 'gloo'})
    assert 'mp_ops_worker ok' in out


@pytest.mark.gpu
def test_rccl_backend_executes_on_one_rank():
    """The shipped transport (backend nccl = RCCL) initialised and used for real
    on the one GPU of the test box: see mp_nccl_worker.py for what that can and
    cannot show."""
    out = _run('mp_nccl_worker.py', 1, {'STK_BACKEND': 'nccl',
                                        'STK_FORCE_COLLECTIVES': '1'})
    assert 'mp_nccl_worker ok' in out


@pytest.mark.gpu
def test_c_abi_communication(tmp_path):
    """stk_comm_* (the neighbour exchange and the scalar all-reduce for
    a host without torch.distributed; RCCL loaded by libstk itself) through
    ctypes: one rank on the test box's GPU (self-addressed messages really travel
    through RCCL), one rank per GPU where two are visible."""
    import torch
    ranks = [1] + ([2] if torch.cuda.device_count() >= 2 else [])
    for n in ranks:
        id_file = str(tmp_path / ('rccl_id_%d' % n))
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r),
                       STK_COMM_ID_FILE=id_file)
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'mp_ccomm_worker.py')],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        for r, pr in enumerate(procs):
            out, _ = pr.communicate(timeout=300)
            assert pr.returncode == 0 and 'mp_ccomm_worker ok' in out, out[-3000:]


@pytest.mark.gpu
def test_distributed_solve_over_rccl_on_two_gpus():
    """The same worker with one rank per GPU and backend nccl (= RCCL over xGMI):
    ghost-row send/recv, wavelet partner exchange and the scalar all-reduce on
    device buffers.  Needs a node with two GPUs; the one-GPU test box skips it
    (the driver's 8-GPU bench is then the first execution of this transport
    across devices)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    out = _run('mp_gpu_worker.py', 2, {'STK_BACKEND': 'nccl',
                                       'STK_TEST_PROBLEM': 'square'})
    assert 'mp_gpu_worker ok' in out
    out = _run('mp_ops_worker.py', 2, {'STK_BACKEND': 'nccl'})
    assert 'mp_ops_worker ok' in out


@pytest.mark.gpu
@pytest.mark.parametrize('nproc,backend', [(1, 'nccl'), (2, 'gloo'), (4, 'gloo')])
def test_bench_runs_as_the_driver_launches_it(nproc, backend):
    """bench.py itself (kron steps + a short solve), launched as the driver
    launches it: one rank on the RCCL-backed communicator, and 2 / 4 ranks sharing
    the test box's GPU over gloo (time slabs with ghost rows, halo exchange every
    step, max-over-ranks timing, one JSON line from rank 0)."""
    import json
    env = dict(os.environ, STK_BACKEND=backend, STK_FORCE_COLLECTIVES='1',
               OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()),
           os.path.join(os.path.dirname(HERE), 'bench.py'), '--gpus', str(nproc),
           '--steps', '2', '--warmup', '1', '--J_time', '4', '--J_space', '5',
           '--solve-iters', '2', '--no-cpu-baseline',
           # several ranks: WITH the untimed preheat of the default flags -- its number of
           # steps (each a halo exchange) must be agreed between the ranks, not read off each
           # rank's own clock (round 6: a two-rank run at config 3 stopped there for good)
           '--preheat', '0.3']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == nproc and rec['value'] > 0 and rec['pcg']['iters_timed'] >= 1
    assert rec['scaling'] == 'strong' and rec['roofline']['frac'] > 0
    assert rec['pcg']['roofline']['bytes_per_iteration'] > 0
    if nproc > 1:
        assert 'ghost' in rec['roofline']['kernel']
        # first-contact record of a multi-rank run: where every rank's step goes, the
        # halo form the start-up probe chose, the ceiling the wire sets, and one
        # start-up line per rank on stderr
        mg = rec['multi_gpu']
        assert [r['rank'] for r in mg['per_rank']] == list(range(nproc))
        for r in mg['per_rank']:
            assert r['step_ms'] > 0 and r['host_wait_for_wire_ms'] >= 0 and r['pack_ms'] > 0
            assert r['pass_without_ghosts_ms'] > 0 and r['ghost_share_ms'] > 0
        assert mg['halo_form']['chosen'] >= 1 and mg['halo_form']['reason']
        if nproc > 2:
            assert mg['halo_form']['identical'] and all(mg['halo_form']['identical'].values())
        assert mg['wire_bound']['ceiling_GBs'] > 0
        # ... and where an ITERATION of the solve goes on every rank: device time of S, P,
        # W, W^T per apply, the host's wait for S's one halo, the scalar all-reduces
        assert [r['rank'] for r in mg['solve_per_rank']] == list(range(nproc))
        for r in mg['solve_per_rank']:
            assert r['S_device_ms_per_apply'] > 0 and r['P_device_ms_per_apply'] > 0
            assert r['W_device_ms_per_apply'] > 0 and r['WT_device_ms_per_apply'] > 0
            assert r['S_host_wait_for_halo_ms_last_apply'] >= 0
            assert r['allreduce_calls_per_iteration'] >= 2 and r['allreduce_host_ms_each'] > 0
        assert res.stderr.count('stk start-up:') == nproc


def test_bench_builds_its_own_launch_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks under
    torch.distributed.run on 127.0.0.1 as CHILD processes (CPU check of the
    command; the GPU test below runs it)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
    assert bench.spawn_ranks(4) == 0
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert '--standalone' in cmd and cmd[cmd.index('--local-addr') + 1] == '127.0.0.1'
    assert cmd[-5:] == [os.path.abspath(bench.__file__), '--gpus', '4', '--steps', '3']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # under another launcher (mpirun sets OMPI_* and no RANK) every process would start
    # N more: refused, as is a start under a profiler preload
    monkeypatch.setenv('OMPI_COMM_WORLD_SIZE', '4')
    with pytest.raises(SystemExit) as exc:
        bench.spawn_ranks(4)
    assert 'another launcher' in str(exc.value)
    monkeypatch.delenv('OMPI_COMM_WORLD_SIZE')
    monkeypatch.setenv('ROCPROFILER_SOMETHING', '1')
    monkeypatch.setenv('LD_PRELOAD', '/opt/rocm/lib/librocprofiler-sdk-tool.so')
    with pytest.raises(SystemExit) as exc:
        bench.spawn_ranks(4)
    assert 'profiler' in str(exc.value)


@pytest.mark.gpu
def test_bench_started_without_a_launcher_spawns_its_ranks():
    """The driver's command shape, `python bench.py --gpus N ...`, for N = 2: the
    bench starts its two ranks itself (here sharing the box's GPU over gloo) and
    rank 0's JSON line comes back on stdout."""
    import json
    env = dict(os.environ, STK_BACKEND='gloo', OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), 'bench.py'), '--gpus', '2',
           '--steps', '2', '--warmup', '1', '--J_time', '4', '--J_space', '5',
           '--solve-iters', '2', '--no-cpu-baseline', '--preheat', '0']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    assert rec['n_gpus'] == 2 and rec['value'] > 0 and 'ghost' in rec['roofline']['kernel']
