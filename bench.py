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

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, 'spacetime-fullgrid-parallel_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def seeded_slab(t_begin, t_end, M):
    """x ~ U[0,1), generated per GLOBAL time row so that every rank count sees
    the same global vector (protocol of reference heateq_mpi_timing.py:81-83,
    which seeds 128)."""
    out = np.empty((t_end - t_begin, M))
    for t in range(t_begin, t_end):
        out[t - t_begin] = np.random.RandomState(128 + t).rand(M)
    return out


def cpu_baseline(A_t, M_t, M_x, A_x, N, M, nbytes):
    """The CPU oracle (NumPy/SciPy restatement of the reference's
    SumMPI([TridiagKronMatMPI, TridiagKronMatMPI]) path) on one host core."""
    from oracle import kron as okron
    rows = min(N, 17)  # bounded sample: the leading 17 time rows
    T1, T2 = A_t[:rows, :rows].tocsr(), M_t[:rows, :rows].tocsr()
    X = seeded_slab(0, rows, M)
    okron.sum_apply([(T1, M_x), (T2, A_x)], X)  # warm
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or time.perf_counter() - t0 < 8.0:
        okron.sum_apply([(T1, M_x), (T2, A_x)], X)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    sample_bytes = 16 * rows * M + 12 * (M_x.nnz + A_x.nnz) + 8 * (M + 1)
    return {
        'value': sample_bytes / dt / 1e9,
        'unit': 'GB/s',
        'cores': 1,
        'kind': 'port',
        'sample': 'leading %d of %d time rows, %d applies, %.3f s each' %
                  (rows, N, reps, dt),
    }


def pmc_traffic(args, size):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane
    streams, plus WRITE_SIZE, both KiB), collected by tools/pmc_kron.sh in
    separate runs and committed under profiles/.  Only meaningful for the
    configuration it was measured on."""
    path = os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')
    if not os.path.exists(path) or size != 1:
        return None
    rec = json.load(open(path))
    if (rec.get('J_time'), rec.get('J_space'), rec.get('problem')) != (
            args.J_time, args.J_space, args.problem):
        return None
    return rec['hbm_bytes_per_launch']


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
    ap.add_argument('--preheat', type=float, default=0.5,
                    help='seconds of untimed applies before the warm-up steps')
    args = ap.parse_args()

    import torch
    from source.comm import MPI
    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    assert size == args.gpus, 'launch one process per GPU (torchrun)'
    assert torch.cuda.is_available(), 'bench.py needs a GPU; no CPU fallback'

    from source.assembly import space_matrices, time_matrices
    from source.mpi_kron import SumMPI, TridiagKronMatMPI
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    from source.problem import problem_helper

    mesh_space, _, mesh_time, data, _ = problem_helper(args.problem,
                                                       J_space=args.J_space,
                                                       J_time=args.J_time)
    A_t, L_t, M_t, G_t, u0_t = time_matrices(mesh_time)
    M_x, A_x = space_matrices(mesh_space)
    N, M = A_t.shape[0], M_x.shape[0]
    dd = DofDistributionMPI(comm, N, M)
    op = SumMPI(dd, [TridiagKronMatMPI(dd, A_t, M_x),
                     TridiagKronMatMPI(dd, M_t, A_x)])
    fused = op._groups[0]
    x = KronVectorMPI(dd, seeded_slab(dd.t_begin, dd.t_end, M))
    y = x._like()
    n_loc = dd.t_end - dd.t_begin
    my_bytes = fused.algorithmic_bytes(n_loc, M)

    def step():
        x._invalidate()  # forces the halo exchange, as heateq_mpi_timing.py:94
        op._matvec(x, y)

    # bring the device to its steady clocks first: a cold GPU runs the first few
    # milliseconds of work measurably slower (the same kernel: 0.40 ms in the
    # first 10 ms after start-up, 0.37 ms later)
    t_hot = time.perf_counter() + args.preheat
    while time.perf_counter() < t_hot:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    comm.Barrier()
    torch.cuda.synchronize()
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
    solve = None
    if args.solve_iters > 0:
        import heateq_mpi as hm
        from source.linalg import PCG
        h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time,
                               problem=args.problem)
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
        solve = {'iters_timed': n_it, 'iters_per_s': n_it / ds,
                 'ms_per_iter': ds / n_it * 1e3,
                 'r_dot_Pr': [float(v) for v in hist]}

    if rank != 0:
        return
    achieved = my_bytes / (kernel_ms * 1e-3) / 1e9
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
            'traffic': pmc_traffic(args, size),
            'kernel': 'kron_ell_kernel<NT=2, shared input, K=7>' + (
                '' if size == 1 else ' + kron_ell_ghost_kernel (rank 0 slab)'),
            'bytes_per_launch': my_bytes,
            'avg_launch_ms': kernel_ms,
            'step_ms_with_halo_exchange': dev_ms,
        },
        'pcg': solve,
    }
    if size == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(A_t, M_t, M_x, A_x, N, M, my_bytes)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
