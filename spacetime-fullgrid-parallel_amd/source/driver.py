"""What the two command-line drivers (heateq_mpi.py, heateq_mpi_timing.py) share:
the command line of the reference's drivers, the lines they print, the
per-operator counters and the gathered record (reference heateq_mpi.py:259-312,
heateq_mpi_timing.py:62-128)."""
import argparse
import base64
import pickle
import sys
import zlib

import numpy as np
import torch

from .comm import MPI

# (flag, type, default, help): the options both reference drivers take
_PROBLEM_OPTIONS = (
    ('problem', str, 'square', 'problem type (square, lshape, cube)'),
    ('J_time', int, 7, 'number of time refines'),
    ('J_space', int, 7, 'number of space refines'),
    ('smoothsteps', int, 3, 'number of smoothing steps'),
    ('vcycles', int, 2, 'number of vcycles'),
    ('wavelettransform', str, 'composite', 'type of wavelettransform'),
    ('alpha', float, 0.3, 'alpha'),
)
OPERATORS = ('W', 'S', 'WT', 'P')


def device_mb():
    """Device memory in use by this process, MB (the reference reports host RSS;
    the vectors and matrices live in HBM here)."""
    return torch.cuda.memory_allocated() / 1048576 if torch.cuda.is_available() else 0.0


def parse(description, argv, extra=(), defaults=None):
    """`defaults`: a driver's own defaults where the reference's drivers differ (its
    timing script takes wavelettransform='original', heateq_mpi_timing.py:35-37, its solve
    driver 'composite', heateq_mpi.py:225-228)."""
    parser = argparse.ArgumentParser(description=description)
    for flag, kind, default, text in _PROBLEM_OPTIONS + tuple(extra):
        parser.add_argument('--' + flag, type=kind, default=(defaults or {}).get(flag, default), help=text)
    return parser.parse_args(argv)


def solver_arguments(args):
    """Keyword arguments of HeatEquationMPI from a parsed command line."""
    keys = [flag for flag, _, _, _ in _PROBLEM_OPTIONS] + ['schur', 'arithmetic']
    return {k: getattr(args, k) for k in keys if hasattr(args, k)}


def start(args):
    """Communicator, rank and size; refuses more ranks than time steps and
    prints the opening lines on rank 0."""
    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    if size > 2**args.J_time + 1:
        print('Too many ranks!')
        sys.exit('1')
    if rank == 0:
        print('\n\nCreating mesh with %d time refines and %d space refines.'
              % (args.J_time, args.J_space))
        print('GPU ranks: %d ' % size)
        print('Arguments: %s' % args)
    return comm, rank, size


def report_construction(heat):
    print('N = %d. M = %d.' % (heat.N, heat.M))
    print('Constructed bilinear forms in %s s.' % heat.setup_time)
    print('Device memory after construction: %smb.' % device_mb())


def counters(op, **more):
    record = {'time_applies': op.time_applies,
              'time_communication': op.time_communication,
              'num_applies': op.num_applies}
    record.update(more)
    return record


def seeded_vector(heat, vector_type, seed=128):
    """The timing driver's input: uniform random numbers, drawn per GLOBAL time
    row (seed + t) so that every rank count sees the same global vector (the
    reference seeds 128 and draws the local block)."""
    dd = heat.dofs_distr
    vec = vector_type(dd)
    for t in range(dd.t_begin, dd.t_end):
        row = np.random.RandomState(seed + t).rand(heat.M)
        vec.X_loc[t - dd.t_begin] = torch.from_numpy(row).to(vec.buf.device)
    return vec


def time_operator(comm, op, vec, iters):
    """`iters` applies of op to vec, the halo re-exchanged each time; one untimed
    apply first (plans, workspaces).  Returns the operator's record."""
    op @ vec
    op.num_applies = op.time_applies = op.time_communication = 0
    began = MPI.Wtime()
    per_apply, per_exchange = [], []
    for _ in range(iters):
        before = (op.time_applies, op.time_communication)
        vec._invalidate()
        op @ vec
        per_apply.append(op.time_applies - before[0])
        per_exchange.append(op.time_communication - before[1])
        comm.Barrier()
    return counters(op, time_applies_iter=per_apply,
                    time_communication_iter=per_exchange,
                    time_total=MPI.Wtime() - began)


def publish(comm, record):
    """Gathers the per-rank records on rank 0 and prints them as the reference's
    `data:` line, base64(zlib(pickle))."""
    everyone = comm.gather(record, root=0)
    if comm.Get_rank() == 0:
        blob = base64.b64encode(zlib.compress(pickle.dumps(everyone)))
        print('\ndata: %s' % str(blob, 'ascii'))
    return everyone
