"""Times the four operators W, S, WT, P separately (counterpart of the
reference's heateq_mpi_timing.py: --iters applies of each operator on a seeded
random vector, vec._invalidate() before each so the halo is re-exchanged).

    python heateq_mpi_timing.py --J_time=6 --J_space=9 --iters 10
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        heateq_mpi_timing.py --J_time=6 --J_space=9
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if __name__ == '__main__':  # as heateq_mpi.py: before anything starts a thread
    from source.host_malloc import keep_to_the_heap
    keep_to_the_heap()

from heateq_mpi import HeatEquationMPI  # noqa: E402
from source import driver  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.mpi_kron import LinearOperatorMPI  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402


def main(argv=None):
    args = driver.parse('Time several components of the parallel heat equation.', argv,
                        extra=[('iters', int, 10, 'number of iterations per operator')],
                        defaults={'wavelettransform': 'original'})  # the reference's default here
    comm, rank, size = driver.start(args)
    heat = HeatEquationMPI(**driver.solver_arguments(args))
    if rank == 0:
        driver.report_construction(heat)

    LinearOperatorMPI.sync_timing = True  # time_applies = device time per apply
    comm.Barrier()
    began = MPI.Wtime()
    vec = driver.seeded_vector(heat, KronVectorMPI)
    record = {'rank': rank}
    for name in driver.OPERATORS:
        record[name] = driver.time_operator(comm, getattr(heat, name), vec, args.iters)
    comm.Barrier()
    record['time_total'] = MPI.Wtime() - began
    record['mem_after_timing'] = driver.device_mb()
    if rank == 0:
        print('\nCompleted %d iters steps.' % args.iters)
        print('Total time: %ss.' % record['time_total'])
        print('      apply      communication   (seconds per apply)')
        heat.print_time_per_apply()
        print('Device memory after timing: %smb.' % record['mem_after_timing'])
    driver.publish(comm, record)
    return record


if __name__ == "__main__":
    main()
