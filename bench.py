#!/usr/bin/env python3
"""Benchmark of the space-time Kronecker hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric (BASELINE.json): throughput of the Kronecker matvec
    y = (A_t kron M_x + M_t kron A_x) x
at J_time = 6, J_space = 9 on the unit square, in GB/s of ALGORITHMIC bytes
(x once, y once, ghost rows, every CSR array once), plus
PCG iterations per second of the preconditioned solve as a secondary figure.
One "step" = one apply of that operator to a resident vector (including the
halo exchange when the time axis is sharded over several GPUs).  The problem
is fixed while GPUs are added (strong scaling): each rank owns one time slab.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, 'spacetime-fullgrid-parallel_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
if __name__ == '__main__':
    # the allocator policy the drivers run under (source/host_malloc.py), set before
    # anything starts a thread; reported in the line as pcg.host_allocator
    from source.host_malloc import keep_to_the_heap
    HOST_ALLOCATOR = 'heap' if keep_to_the_heap() else 'default'
else:
    HOST_ALLOCATOR = 'default'

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
XGMI_LINK_GBS_PER_DIRECTION = 76.8  # xGMI: 7 point-to-point links per GPU, ~153.6 GB/s bidirectional each


def seeded_slab(t_begin, t_end, M):
    """Rows [t_begin, t_end) of ONE global vector `np.random.RandomState(128).rand(N, M)`
    (drawn globally and then sliced, so that every rank count
    sees the same data).  On one rank this is the reference's timing vector
    (heateq_mpi_timing.py:81-83); on several ranks it deliberately DEVIATES from the
    reference, which seeds 128 on every rank and draws rand(*X_loc.shape) locally, so
    that all its ranks hold the same leading rows.  The generator's stream is
    row-major: the rows before the slab are drawn and dropped one at a time."""
    rs = np.random.RandomState(128)
    for _ in range(t_begin):
        rs.rand(M)
    return rs.rand(t_end - t_begin, M)


_CPU = {}


def _cpu_slab(args):
    """One worker of the all-cores baseline: `reps` applies of the time rows
    [t0, t1) of y = (A_t kron M_x + M_t kron A_x) x through the oracle's
    functions (T_t kron I first, then I kron X_x; reference mpi_kron.py:214-219).
    The matrices are inherited from the parent (fork)."""
    t0, t1, reps = args
    from oracle import kron as okron
    A_t, M_t, M_x, A_x, N, M = (_CPU[k] for k in ('A_t', 'M_t', 'M_x', 'A_x', 'N', 'M'))
    lo, hi = max(t0 - 1, 0), min(t1 + 1, N)
    X = seeded_slab(lo, hi, M)  # the rows the slab's time stencil reaches
    T1, T2 = A_t[t0:t1, lo:hi].tocsr(), M_t[t0:t1, lo:hi].tocsr()
    tic = time.perf_counter()
    for _ in range(reps):
        out = okron.identity_kron_mat(M_x, okron.tridiag_kron_identity(T1, X))
        out += okron.identity_kron_mat(A_x, okron.tridiag_kron_identity(T2, X))
    return time.perf_counter() - tic


def cpu_baseline(A_t, M_t, M_x, A_x, N, M, nbytes, budget_s=10.0):
    """The CPU oracle (NumPy/SciPy restatement of the reference's
    SumMPI([TridiagKronMatMPI, TridiagKronMatMPI]) path, kind 'port') on the
    host cores of this box, protocol of reference heateq_mpi_timing.py:81-102
    (seeded vector, warm-up, repeated applies): the WHOLE N time rows on one
    core, and on all cores as independent time slabs in a process pool
    (SURVEY.md section 8d).  Runs in a process that never touches the GPU (bench.py
    starts itself with --cpu-baseline-only as a child), so the pool can fork."""
    import multiprocessing as mp
    import platform
    from oracle import kron as okron
    terms = [(A_t, M_x), (M_t, A_x)]
    X = seeded_slab(0, N, M)
    okron.sum_apply(terms, X)  # warm
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or time.perf_counter() - t0 < budget_s:
        okron.sum_apply(terms, X)
        reps += 1
    dt1 = (time.perf_counter() - t0) / reps
    del X
    cores = len(os.sched_getaffinity(0))
    _CPU.update(A_t=A_t.tocsr(), M_t=M_t.tocsr(), M_x=M_x, A_x=A_x, N=N, M=M)
    workers = min(cores, N)
    edges = np.linspace(0, N, workers + 1).astype(int)
    # as many applies per worker as fill the budget at the one-core rate
    preps = max(2, int(budget_s / max(dt1 / workers, 1e-3)))
    with mp.get_context('fork').Pool(workers) as pool:
        pool.map(_cpu_slab, [(int(a), int(b), 1) for a, b in zip(edges[:-1], edges[1:])])  # warm
        tic = time.perf_counter()
        pool.map(_cpu_slab, [(int(a), int(b), preps) for a, b in zip(edges[:-1], edges[1:])])
        dtp = (time.perf_counter() - tic) / preps
    model = platform.processor() or ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {
        'value': nbytes / dtp / 1e9,
        'unit': 'GB/s',
        'cores': workers,
        'kind': 'port',
        'one_core_value': nbytes / dt1 / 1e9,
        'cpu_model': model,
        'sample': 'all %d time rows; 1 core: %d applies of %.3f s; %d cores '
                  '(independent time slabs, process pool): %d applies of %.3f s'
                  % (N, reps, dt1, workers, preps, dtp),
    }


def pcg_cpu_baseline(problem, J_time, J_space):
    """ONE iteration of PCG(W^T S W, P, rhs) through the CPU oracle -- the loop the
    reference times (heateq_mpi.py:281-288; operators wired as heateq_mpi.py:166-185:
    five-term S with four multigrid applies, per-level wavelet steps, block-diagonal
    P) -- on the host cores of this box: time slices are independent in every space
    operator, so batches of slices run side by side on all cores
    (oracle.multigrid.THREADS; the sweeps themselves are the C restatement
    oracle/gs.c, the rest SciPy).  The iteration timed is the first one (from
    `t = T p` to the update of p); one iteration is the whole sample: at config 3
    it is tens of seconds of CPU work."""
    from oracle import multigrid as omg
    from oracle.heat import HeatEquationOracle
    from oracle.krylov import _dot
    from source.assembly import prolongation_matrices, space_load, space_matrices, time_matrices
    from source.problem import problem_helper
    t_setup = time.perf_counter()
    mesh, _, tmesh, data, _ = problem_helper(problem, J_space, J_time)
    A_t, L_t, M_t, G_t, u0_t = time_matrices(tmesh)
    M_x, A_x = space_matrices(mesh, scipy_path=True)
    o = HeatEquationOracle(dict(A_t=A_t, L_t=L_t, M_t=M_t, G_t=G_t, M_x=M_x, A_x=A_x,
                                P_mats=prolongation_matrices(mesh), u0_t=u0_t,
                                u0_x=space_load(mesh, data['u0'], numpy_path=True)), J_time)
    # threads actually used: batches of time slices are cut into at most N chunks
    cores = min(len(os.sched_getaffinity(0)), o.N)
    omg.THREADS = cores
    b = o.rhs()
    t_setup = time.perf_counter() - t_setup
    w = np.zeros_like(b)
    r = b.copy()
    p = o.P(r)
    abs_r = _dot(r, p)
    tic = time.perf_counter()
    t = o.WT_S_W(p)  # linalg.py:28-41
    alpha = abs_r / _dot(p, t)
    w += alpha * p
    r -= alpha * t
    z = o.P(r)
    abs_r_new = _dot(r, z)
    p *= abs_r_new / abs_r
    p += z
    dt = time.perf_counter() - tic
    return {'value': 1.0 / dt, 'unit': 'iterations/s', 'cores': cores, 'kind': 'port',
            's_per_iteration': dt, 'oracle_setup_s': t_setup,
            'r_dot_Pr': [abs_r, abs_r_new],
            'sample': 'iteration 1 of PCG(W^T S W, P, rhs), J_time=%d J_space=%d %s, all %d time '
                      'rows, %d threads over independent time slices; one iteration = %.1f s'
                      % (J_time, J_space, problem, o.N, cores, dt)}


KERNEL_SOURCES = ('csrc/kron_pack.hip', 'csrc/stk_common.h')


def kernel_source_sha():
    """Identity of the CODE the PMC traffic figure belongs to: a hash of the
    sources the benched kernel is compiled from (the kernel file and the header
    it includes -- nothing else: round 4 hashed the whole of source/linop.py and
    lost its record to an unrelated edit two minutes after the PMC pass)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(PKG, rel), 'rb').read())
    return h.hexdigest()[:16]


def plan_sha(packed):
    """Identity of the DATA the kernel streams: a hash of the packed plan itself
    (slot words, row list, dictionary, the sizes that fix the launch), whichever
    planner built it.  Together with kernel_source_sha() this names what a PMC
    record was measured on; an edit that changes neither cannot change the
    kernel's traffic."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.array([packed.M, packed.K, packed.col_bits, packed.n_codes, packed.n_mats,
                       packed.rows_per_unit, packed.n_units], dtype=np.int64).tobytes())
    for arr in (packed.slots, packed.row_ids, packed.dict):
        if arr is not None:
            h.update(np.ascontiguousarray(arr.cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


def pmc_traffic(args, size, kernel, plan):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes
    (tools/pmc_passes.sh + tools/pmc_traffic.py: FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE, both KiB;
    separate passes).  The record names the kernel, the hash of the kernel's
    sources and the hash of the plan it was measured on; a record of another
    kernel, code, plan or configuration is refused (null), not reported."""
    import glob
    if size != 1:
        return None, None
    sha = kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        rec = json.load(open(path))
        if ((rec.get('J_time'), rec.get('J_space'), rec.get('problem')) == (
                args.J_time, args.J_space, args.problem)
                and rec.get('source_sha') == sha and rec.get('plan_sha') == plan
                and rec.get('kernel', '') in kernel):
            return rec['hbm_bytes_per_launch'], {
                'file': os.path.relpath(path, REPO), 'kernel': rec['kernel'],
                'source_sha': sha, 'plan_sha': plan, 'fetch_size_kib': rec.get('fetch_size_kib'),
                'write_size_kib': rec.get('write_size_kib')}
    return None, {'source_sha': sha, 'plan_sha': plan, 'note': 'no PMC record for this code and plan'}


def pcg_byte_model(h, n_loc):
    """Algorithmic bytes of ONE iteration of PCG(W^T S W, P, rhs) on a slab of
    n_loc time rows, from the operator list of SURVEY.md section 3.1: every
    operator moves its vector operands once (8 bytes per entry) and its matrix
    once in the reference's CSR format (12 bytes per entry + 4 per row); level
    sizes and entry counts are those of the Galerkin hierarchies actually built.
    Returns (bytes of the decomposition this build runs, breakdown); the
    breakdown also carries the total of the reference's own decomposition
    (five-term S with four multigrid applies, per-level wavelet steps)."""
    from source.wavelets import WaveletTransformOp
    N = n_loc

    def V(n):
        return 8.0 * N * n

    def csr(m):
        return 12.0 * m.nnz + 4.0 * (m.shape[0] + 1)

    def mg_apply(mats, P_mats, smoothsteps, vcycles, tight=False):
        cyc = 2.0 * V(mats[0].shape[0])  # exact solve on level 0
        for l in range(1, len(mats)):
            n, nc = mats[l].shape[0], mats[l - 1].shape[0]
            sweep = 3.0 * V(n) + csr(mats[l])
            # tight: a smoothing call streams u, f and the matrix ONCE however many
            # sweeps it runs (what a sweep order that kept its operands on chip
            # would need); otherwise every sweep is a full pass, which is what the
            # reference's sweep order costs this memory system
            cyc += 2 * (1 if tight else smoothsteps) * sweep  # pre- and post-smoothing
            cyc += 3.0 * V(n) + csr(mats[l])        # residual
            cyc += V(n) + V(nc) + csr(P_mats[l - 1])  # restriction
            cyc += V(nc)                            # zero coarse iterate
            cyc += V(nc) + 2.0 * V(n) + csr(P_mats[l - 1])  # prolongate + correct
        return vcycles * cyc + V(mats[-1].shape[0])

    M = h.M
    P_mats = h.hierarchy.P_mats
    kd, fd = h.Kinv_x._dev, h.C_family._dev
    mg_K = mg_apply(kd.mats_a, P_mats, kd.smoothsteps, kd.vcycles)
    # the preconditioner's matrices 2^j M + alpha A live on the union pattern
    union = [sp_union(a, m) for a, m in zip(fd.mats_a, fd.mats_m)]
    mg_C = mg_apply(union, P_mats, fd.smoothsteps, fd.vcycles)
    mg_K_tight = mg_apply(kd.mats_a, P_mats, kd.smoothsteps, kd.vcycles, tight=True)
    mg_C_tight = mg_apply(union, P_mats, fd.smoothsteps, fd.vcycles, tight=True)
    spmv_M, spmv_A = 2.0 * V(M) + csr(h.M_x), 2.0 * V(M) + csr(h.A_x)
    tri = 2.0 * V(M)
    S = (4 * (tri + mg_K) + (spmv_M + spmv_M) + (spmv_M + spmv_A) + (spmv_A + spmv_M)
         + (spmv_A + spmv_A) + (tri + spmv_M) + 4 * 3.0 * V(M))
    wt = WaveletTransformOp(h.J_time, interleaved=True)
    W = 0.0
    for j in range(1, h.J_time + 1):
        W += 2.0 * V(M) + 24.0 * M * wt.split(j).nnz * N / float(h.N)
    P = 2.0 * mg_C + spmv_A
    blas1 = 15.0 * V(M)
    reference = 2.0 * W + S + P + blas1
    # the decomposition actually run: S regrouped so that K
    # is applied twice instead of four times around three fused Kronecker passes
    # (SchurMPI), W and W^T as one fused pass each; multigrid, P and BLAS-1 as above
    cMA = csr(h.M_x) + csr(h.A_x)
    S_impl = 2.0 * mg_K + 2.0 * (2.0 * V(M) + cMA) + (4.0 * V(M) + cMA)
    W_impl = 2.0 * V(M)
    total = 2.0 * W_impl + S_impl + P + blas1
    tight = total - 2.0 * (mg_K - mg_K_tight) - 2.0 * (mg_C - mg_C_tight)
    return total, {'S': S_impl, 'W_and_WT': 2.0 * W_impl, 'P': P, 'blas1': blas1,
                   'mg_apply_K': mg_K, 'mg_apply_C': mg_C,
                   'mg_apply_K_tight': mg_K_tight, 'mg_apply_C_tight': mg_C_tight,
                   'total_tight': tight,
                   'reference_decomposition_total': reference,
                   'reference_decomposition_S': S, 'reference_decomposition_W_and_WT': 2.0 * W}


def sp_union(a, m):
    import scipy.sparse as sp
    u = sp.csr_matrix(abs(a) + abs(m))
    return u


def profiler_preloaded():
    """True under rocprofv3 / rocprof: their preloaded tool library initialises
    the GPU before the program starts (it does with --pmc)."""
    pre = os.environ.get('LD_PRELOAD', '')
    return ('rocprof' in pre or 'roctracer' in pre
            or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ))


EXTERNAL_LAUNCHERS = ('OMPI_COMM_WORLD_SIZE', 'PMI_SIZE', 'PMIX_RANK', 'SLURM_NTASKS')


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N ranks as child processes
    of torch.distributed.run on 127.0.0.1, a free port; returns their exit code.
    Refused under another launcher (mpirun / srun start N processes themselves:
    each would start N more) and under a profiler whose preloaded library has
    already initialised the GPU in this process."""
    import subprocess
    found = [k for k in EXTERNAL_LAUNCHERS if k in os.environ]
    if found:
        sys.exit('bench.py --gpus %d: started under another launcher (%s) without RANK / '
                 'WORLD_SIZE; launch it with torch.distributed.run, one process per GPU'
                 % (n, ', '.join(found)))
    if profiler_preloaded():
        sys.exit('bench.py --gpus %d: a profiler preload has initialised the GPU; put the '
                 'profiler around torch.distributed.run instead' % n)
    # --standalone: the launcher binds a free port itself (no probe-then-bind window)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr',
           '127.0.0.1', '--nnodes=1', '--nproc-per-node', str(n),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--J_time', type=int, default=6)
    ap.add_argument('--J_space', type=int, default=9)
    ap.add_argument('--problem', default='square')
    ap.add_argument('--solve-iters', type=int, default=10,
                    help='PCG iterations to time for iters/s (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=10.0,
                    help='time budget of each leg (1 core, all cores) of the CPU baseline')
    ap.add_argument('--preheat', type=float, default=0.5,
                    help='seconds of untimed applies before the warm-up steps')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='(internal) run the CPU baseline alone and print its record as JSON')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # started as plain `python bench.py --gpus N` (the reference's protocol is
        # `mpirun -np N`, README.md:26-31): start the N ranks ourselves, one
        # process per GPU, as children under torch.distributed.run -- before this
        # process has touched the GPU, and never by exec.  Rank 0's JSON line
        # arrives on the inherited stdout.
        sys.exit(spawn_ranks(args.gpus))

    from source.assembly import space_matrices, time_matrices
    from source.problem import problem_helper

    mesh_space, _, mesh_time, data, _ = problem_helper(args.problem,
                                                       J_space=args.J_space,
                                                       J_time=args.J_time)
    A_t, L_t, M_t, G_t, u0_t = time_matrices(mesh_time)
    # (the SciPy form of the assembly: the same matrices bit for bit on these meshes,
    # and libstk -- with the HIP runtime it links -- stays unloaded until the CPU
    # baseline's process pool has forked and finished)
    M_x, A_x = space_matrices(mesh_space, scipy_path=True)
    N, M = A_t.shape[0], M_x.shape[0]

    # The CPU baseline (rank 0 of a one-GPU run only) runs in a CHILD process, and LAST:
    # see run_cpu_baseline below.
    cpu = None
    if args.cpu_baseline_only:
        nb = 16 * N * M + 12 * (M_x.nnz + A_x.nnz) + 8 * (M + 1)
        rec = cpu_baseline(A_t, M_t, M_x, A_x, N, M, nb, args.cpu_seconds)
        if args.solve_iters > 0:
            rec['pcg'] = pcg_cpu_baseline(args.problem, args.J_time, args.J_space)
            rec['pcg']['cpu_model'] = rec['cpu_model']
        print(json.dumps(rec))
        return
    if profiler_preloaded() and not args.no_cpu_baseline:
        # under a profiler the program must not start other processes
        args.no_cpu_baseline = True
        print('bench.py: profiler preload detected, CPU baseline skipped', file=sys.stderr)

    def run_cpu_baseline():
        """The baseline in a process of its own, started as a child AFTER every GPU
        measurement of this run is complete.  On some boxes of the pool the Kronecker
        kernel runs 6 % slower for many seconds after all host cores have been busy --
        0.291-0.296 ms in runs that had the baseline before the GPU part, 0.273-0.275 ms
        in runs without it, whichever process ran the baseline
        (profiles/r04_bench_baseline_in_process.log); other boxes do not care.  The
        timed GPU region should not depend on what this script did to the host before."""
        import subprocess
        child = subprocess.run(
            [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--J_time', str(args.J_time),
             '--J_space', str(args.J_space), '--problem', args.problem, '--cpu-seconds', str(args.cpu_seconds),
             '--solve-iters', str(args.solve_iters)],
            capture_output=True, text=True)
        lines = [ln for ln in child.stdout.splitlines() if ln.startswith('{')]
        if child.returncode != 0 or not lines:
            sys.exit('bench.py: the CPU baseline failed:\n' + child.stdout[-2000:] + child.stderr[-2000:])
        return json.loads(lines[-1])

    import torch
    from source.comm import MPI
    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    assert size == args.gpus, 'launch one process per GPU (torchrun)'
    assert torch.cuda.is_available(), 'bench.py needs a GPU; no CPU fallback'

    from source.mpi_kron import SumMPI, TridiagKronMatMPI
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI

    dd = DofDistributionMPI(comm, N, M)
    op = SumMPI(dd, [TridiagKronMatMPI(dd, A_t, M_x),
                     TridiagKronMatMPI(dd, M_t, A_x)])
    fused = op._groups[0]
    x = KronVectorMPI(dd, seeded_slab(dd.t_begin, dd.t_end, M))
    y = x._like()
    n_loc = dd.t_end - dd.t_begin
    my_bytes = fused.algorithmic_bytes(n_loc, M)

    halo = None
    if size > 1:
        # first contact with a multi-GPU node: every rank checks that its plans live on
        # its own device and reports backend / RCCL / peer access on stderr; the halo
        # form is the direct exchange unless STK_HALO_ROUTES pins a routed one or asks
        # for the probe on real rows (=auto; mpi_vector.choose_halo_form)
        from source.mpi_vector import choose_halo_form, startup_report
        pk = fused.ell.packed_for(n_loc) if getattr(fused, 'use_ell', False) else None
        startup_report(dd, [x.buf, y.buf] + list(fused.tri) +
                       ([pk.slots, pk.dict] if pk is not None and pk.ok else []))
        halo = choose_halo_form(dd)
    wire_wait = [0.0]

    def step():
        x._invalidate()  # forces the halo exchange, as heateq_mpi_timing.py:94
        op._matvec(x, y)
        wire_wait[0] += op.time_communication

    # bring the device to its steady clocks first: a cold GPU runs the first few
    # milliseconds of work measurably slower (the same kernel: 0.40 ms in the
    # first 10 ms after start-up, 0.37 ms later)
    # (several ranks: every step is a halo exchange, so the NUMBER of steps must be the same
    # on all of them -- a clock read per rank is not: the ranks agree after every batch of 20
    # whether the time is up.  Round 6: until then each rank looped on its own clock, and a run
    # on two ranks at this size stopped in the first exchange one rank had and the other had not.)
    t_hot = time.perf_counter() + args.preheat
    while args.preheat > 0:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        if comm.allreduce(1.0 if time.perf_counter() >= t_hot else 0.0) > 0.0:
            break
    for _ in range(args.warmup):
        step()
    comm.Barrier()
    torch.cuda.synchronize()
    wire_wait[0] = 0.0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    comm.Barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps

    # kernels alone (several ranks): the same applies with the halo already
    # present, so that the roofline figure of the kernel excludes the exchange
    if size > 1:
        op._matvec(x, y)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.steps):
            op._matvec(x, y)  # x not invalidated: ghost rows are cached
        ev1.record()
        torch.cuda.synchronize()
        kernel_ms = ev0.elapsed_time(ev1) / args.steps
    else:
        kernel_ms = dev_ms

    # several ranks: where each rank's step goes -- the host's wait for the wire, and
    # the device time of the pieces around it, each timed alone (the received rows
    # are there) -- gathered on rank 0
    per_rank = None
    if size > 1:
        mine = {'rank': rank, 'n_loc': n_loc, 'step_ms': dev_ms, 'kernels_with_cached_halo_ms': kernel_ms,
                'host_wait_for_wire_ms': wire_wait[0] / args.steps * 1e3}
        mine.update(fused.phase_times(x, y) or {})
        per_rank = comm.gather(mine)

    import torch.distributed as dist
    red_dev = comm._device() if size > 1 else 'cuda'  # nccl: device, gloo (tests): host
    t = torch.tensor([dt, float(my_bytes)], dtype=torch.float64, device=red_dev)
    if size > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, total_bytes = float(tmax[0]), float(t[1])
    else:
        total_bytes = float(my_bytes)
    ms_per_step = dt / args.steps * 1e3
    value = total_bytes / (ms_per_step * 1e-3) / 1e9

    # ---- secondary figure: PCG iterations per second --------------------------
    # in the library's default arithmetic ('accurate': every r.Pr within 1e-10 of the
    # CPU path) and in the 'fast' one (4 % less time, entries within 4.6e-10)
    def timed_solve(arithmetic):
        import heateq_mpi as hm
        from source.linalg import PCG
        h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time,
                               problem=args.problem, arithmetic=arithmetic)
        PCG(h.WT_S_W, h.P, h.rhs, kmax=8)  # warm-up: plans, workspaces, steady clocks
        comm.Barrier()
        hist, stamps = [], []

        def cb(w, r, k):
            # every iteration ends with a host read of r.Pr anyway; the extra
            # synchronisation here only pins the time stamp
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())

        _, its = PCG(h.WT_S_W, h.P, h.rhs, kmax=args.solve_iters + 2,
                     history=hist, callback=cb)
        comm.Barrier()
        # stamps[k] is taken inside iteration k+1, after its operator apply:
        # consecutive stamps are exactly one PCG iteration apart
        n_it = len(stamps) - 1
        ds = torch.tensor([stamps[-1] - stamps[0]], dtype=torch.float64,
                          device=red_dev)
        if size > 1:
            dist.all_reduce(ds, op=dist.ReduceOp.MAX)
        ds = float(ds[0])
        phases = None
        if size > 1:
            # where an iteration goes on THIS rank, in a short pass of its own after the
            # timed one (every apply then ends with an event synchronisation): device
            # time of S / P / W / W^T per apply, the host's wait for the one halo of
            # an S apply, and the scalar all-reduces -- how many per iteration and what
            # one costs the host -- so that a SCALE record splits into compute, wire and
            # latency without a profiler
            from source.mpi_kron import LinearOperatorMPI
            ops = {'S': h.S, 'P': h.P, 'W': h.W, 'WT': h.WT}
            for o in ops.values():
                o.num_applies = o.time_applies = o.time_communication = 0
            LinearOperatorMPI.sync_timing, type(comm).timing = True, True
            comm.reset_counters()
            try:
                _, its3 = PCG(h.WT_S_W, h.P, h.rhs, kmax=4)
            finally:
                LinearOperatorMPI.sync_timing, type(comm).timing = False, False
            phases = {'rank': rank, 'n_loc': n_loc, 'iterations': its3,
                      'allreduce_calls_per_iteration': comm.allreduce_calls / max(its3, 1),
                      'allreduce_host_ms_each': comm.allreduce_host_s / max(comm.allreduce_calls, 1) * 1e3}
            for name, o in ops.items():
                phases[name + '_device_ms_per_apply'] = o.time_applies / max(o.num_applies, 1) * 1e3
            phases['S_host_wait_for_halo_ms_last_apply'] = h.S.time_communication * 1e3
            phases = comm.gather(phases)
        model_bytes, parts = pcg_byte_model(h, n_loc)
        mb = torch.tensor([model_bytes, parts['total_tight']], dtype=torch.float64, device=red_dev)
        if size > 1:
            dist.all_reduce(mb, op=dist.ReduceOp.SUM)
        model_total, model_tight = float(mb[0]), float(mb[1])
        return {'arithmetic': arithmetic, 'setup_s': h.setup_time, 'host_allocator': HOST_ALLOCATOR,
                # when each stage of the set-up was done, and what ran in side threads (begin, end)
                'setup_stages_s': {label: round(at, 3) for label, at in h.setup_timeline},
                'setup_threads_s': {label: [round(b, 3), round(e, 3)] for label, b, e in h.setup_threads},
                'per_rank': phases,
                'iters_timed': n_it, 'iters_per_s': n_it / ds,
                'ms_per_iter': ds / n_it * 1e3,
                'r_dot_Pr': [float(v) for v in hist],
                # secondary roofline: algorithmic bytes of one iteration by the
                # operator list of SURVEY.md section 3.1
                'roofline': {
                    'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS * size,
                    'bytes_per_iteration': model_total,
                    'achieved': model_total / (ds / n_it) / 1e9,
                    'frac': model_total / (ds / n_it) / 1e9 / (HBM_PEAK_GBS * size),
                    # the tight model: every smoothing call streams its operands once
                    # (3 vector passes + the matrix per call instead of per sweep)
                    'bytes_per_iteration_tight': model_tight,
                    'frac_tight': model_tight / (ds / n_it) / 1e9 / (HBM_PEAK_GBS * size),
                    'breakdown_rank0_bytes': parts}}

    solve = solve_fast = solve_per_rank = None
    if args.solve_iters > 0:
        solve = timed_solve('accurate')
        solve_per_rank = solve.pop('per_rank', None)
        torch.cuda.empty_cache()
        solve_fast = timed_solve('fast')
        del solve_fast['roofline']['breakdown_rank0_bytes']
        solve_fast.pop('per_rank', None)

    if rank != 0:
        return
    if args.gpus == 1 and size == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize()
        cpu = run_cpu_baseline()
    achieved = my_bytes / (kernel_ms * 1e-3) / 1e9
    kernel = fused.kernel_name(n_loc)
    pk = fused.ell.packed_for(n_loc) if getattr(fused, 'use_ell', False) and type(fused).use_pack else None
    traffic, traffic_src = pmc_traffic(args, size, kernel, plan_sha(pk) if pk is not None and pk.ok else None)
    out = {
        'metric': 'Kronecker-matvec GB/s (algorithmic bytes; share of 8 TB/s HBM '
                  'peak in roofline.frac) + PCG iters/s, J_time=%d J_space=%d %s'
                  % (args.J_time, args.J_space, args.problem),
        'value': value,
        'unit': 'GB/s',
        'n_gpus': size,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': 'y = (A_t kron M_x + M_t kron A_x) x, P1 on the uniform '
                        'interval x P1 on the red-refined %s, x ~ U[0,1) seed '
                        '128' % args.problem,
            'J_time': args.J_time, 'J_space': args.J_space, 'N': N, 'M': M,
            'nnz_M_x': int(M_x.nnz), 'nnz_A_x': int(A_x.nnz),
            'parallelism': 'time-slab x%d' % size,
            'algorithmic_bytes_total': total_bytes,
        },
        'roofline': {
            'bound': 'hbm',
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic,
            'traffic_source': traffic_src,
            'kernel': kernel,
            'bytes_per_launch': my_bytes,
            'avg_launch_ms': kernel_ms,
            'step_ms_with_halo_exchange': dev_ms,
        },
        'pcg': solve,
        'pcg_fast': solve_fast,
    }
    if size > 1:
        # what the wire allows: x is invalidated every step (the reference's timing
        # protocol), so each step moves one time row of 8 M bytes to and from each
        # neighbour over ONE xGMI link (7 links x ~153 GB/s bidirectional per GPU,
        # ~77 GB/s per direction); routed over k links, a piece of 1 / k of the row
        # travels in each of two phases
        link = XGMI_LINK_GBS_PER_DIRECTION
        k = max(1, int(halo['chosen']))
        wire_ms = 8.0 * M / (link * 1e9) * 1e3 * (1.0 if k == 1 else 2.0 / k)
        out['multi_gpu'] = {
            'halo_form': halo,
            'per_rank': per_rank,
            'solve_per_rank': solve_per_rank,
            'wire_bound': {
                'bytes_per_neighbour_per_step': 8 * M,
                'xgmi_GBs_per_link_and_direction': link,
                'wire_ms_per_step_at_link_rate': wire_ms,
                'ceiling_GBs': total_bytes / (wire_ms * 1e-3) / 1e9,
                'note': 'the micro-benchmark invalidates x every step (heateq_mpi_timing.py:94): '
                        'a step cannot be shorter than the wire time of one time row per '
                        'neighbour, whatever the kernels do; the solve exchanges one row per S apply',
            },
        }
    if cpu is not None:
        pcg_cpu = cpu.pop('pcg', None)
        if pcg_cpu is not None and solve is not None:
            # the same loop on the host: the oracle's first r.Pr entries are the check that
            # both sides iterate on the same problem
            solve['cpu_baseline'] = pcg_cpu
        out['cpu_baseline'] = cpu
    print(json.dumps(out))


if __name__ == '__main__':
    main()
