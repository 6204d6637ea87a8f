"""Times the four operators W, S, WT, P separately (counterpart of the
reference's heateq_mpi_timing.py: --iters applies of each operator on a seeded
random vector, vec._invalidate() before each so the halo is re-exchanged).

    python heateq_mpi_timing.py --J_time=6 --J_space=9 --iters 10
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        heateq_mpi_timing.py --J_time=6 --J_space=9
"""
import argparse
import base64
import os
import pickle
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from heateq_mpi import HeatEquationMPI, mem  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.mpi_kron import LinearOperatorMPI  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser(
        description="Time several components of the parallel heat equation.")
    parser.add_argument('--problem', default='square')
    parser.add_argument('--J_time', type=int, default=7)
    parser.add_argument('--J_space', type=int, default=7)
    parser.add_argument('--smoothsteps', type=int, default=3)
    parser.add_argument('--vcycles', type=int, default=2)
    parser.add_argument('--wavelettransform', default='composite')
    parser.add_argument('--alpha', type=float, default=0.3)
    parser.add_argument('--iters', type=int, default=10,
                        help='number of iterations per operator')
    args = parser.parse_args(argv)

    comm = MPI.COMM_WORLD
    rank, size = comm.Get_rank(), comm.Get_size()
    if size > 2**args.J_time + 1:
        print('Too many ranks!')
        sys.exit('1')
    heat_eq_mpi = HeatEquationMPI(J_space=args.J_space, J_time=args.J_time,
                                  problem=args.problem,
                                  smoothsteps=args.smoothsteps,
                                  vcycles=args.vcycles, alpha=args.alpha,
                                  wavelettransform=args.wavelettransform)
    if rank == 0:
        print('\n\nCreating mesh with {} time refines and {} space refines.'.
              format(args.J_time, args.J_space))
        print('GPU ranks: ', size)
        print('Arguments:', args)
        print('N = {}. M = {}.'.format(heat_eq_mpi.N, heat_eq_mpi.M))
        print('Constructed bilinear forms in {} s.'.format(
            heat_eq_mpi.setup_time))
        print('Device memory after construction: {}mb.'.format(mem()))

    LinearOperatorMPI.sync_timing = True  # time_applies = wall time per apply
    comm.Barrier()
    time_total = MPI.Wtime()
    dd = heat_eq_mpi.dofs_distr
    vec = KronVectorMPI(dd)
    # the reference seeds 128 and draws the local block; draw per global time
    # row instead so every rank count sees the same global vector
    for t in range(dd.t_begin, dd.t_end):
        vec.X_loc[t - dd.t_begin] = torch.from_numpy(
            np.random.RandomState(128 + t).rand(heat_eq_mpi.M)).to(
                vec.buf.device)
    data = {'rank': rank}
    for name, op in [('W', heat_eq_mpi.W), ('S', heat_eq_mpi.S),
                     ('WT', heat_eq_mpi.WT), ('P', heat_eq_mpi.P)]:
        op @ vec  # warm-up (plans, workspaces); not counted
        op.num_applies, op.time_applies, op.time_communication = 0, 0, 0
        time_total_op = MPI.Wtime()
        time_applies_iter, time_communication_iter = [], []
        for _ in range(args.iters):
            t_a, t_c = op.time_applies, op.time_communication
            vec._invalidate()
            op @ vec
            time_applies_iter.append(op.time_applies - t_a)
            time_communication_iter.append(op.time_communication - t_c)
            comm.Barrier()
        # the per-operator record of reference heateq_mpi_timing.py:104-111
        data[name] = {
            'time_applies': op.time_applies,
            'time_communication': op.time_communication,
            'time_applies_iter': time_applies_iter,
            'time_communication_iter': time_communication_iter,
            'num_applies': op.num_applies,
            'time_total': MPI.Wtime() - time_total_op
        }
    comm.Barrier()
    data['time_total'] = MPI.Wtime() - time_total
    data['mem_after_timing'] = mem()
    if rank == 0:
        print('')
        print('Completed {} iters steps.'.format(args.iters))
        print('Total time: {}s.'.format(data['time_total']))
        print('      apply      communication   (seconds per apply)')
        heat_eq_mpi.print_time_per_apply()
        print('Device memory after timing: {}mb.'.format(mem()))
    gathered = comm.gather(data, root=0)
    if rank == 0:
        print('\ndata: {}'.format(
            str(base64.b64encode(zlib.compress(pickle.dumps(gathered))),
                'ascii')))
    return data


if __name__ == "__main__":
    main()
