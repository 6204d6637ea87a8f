#!/usr/bin/env python3
"""Page faults of the first and of later set-ups of a process under the drivers' allocator
policy, and the box's transparent-huge-page settings: is the first set-up's extra time
(0.82 against 0.63 s in the bench line) the first touch of the heap's pages?"""
import os
import sys
import time
import resource

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source.host_malloc import keep_to_the_heap  # noqa: E402
keep_to_the_heap()
import torch  # noqa: E402
import heateq_mpi as hm  # noqa: E402
from source.assembly import space_matrices  # noqa: E402
from source.problem import problem_helper  # noqa: E402

for name in ('enabled', 'defrag', 'shmem_enabled'):
    try:
        print('transparent_hugepage/%s: %s' % (name, open('/sys/kernel/mm/transparent_hugepage/' + name).read().strip()))
    except OSError as err:
        print(name, err)
torch.zeros(1, device='cuda')
# what bench.py has done before its first set-up: a mesh, its matrices, kernels launched
J_space = int(sys.argv[1]) if len(sys.argv) > 1 else 9
J_time = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mesh = problem_helper('square', J_space=J_space, J_time=J_time)[0]
space_matrices(mesh)


def usage():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_minflt, r.ru_majflt, r.ru_utime, r.ru_stime


for rep in range(reps):
    before, t = usage(), time.time()
    h = hm.HeatEquationMPI(J_space=J_space, J_time=J_time)
    torch.cuda.synchronize()
    dt, after = time.time() - t, usage()
    rss = int(open('/proc/self/statm').read().split()[1]) * 4096 / 1e9
    print('set-up %d: %.2f s wall, %d minor faults, user %.2f s, system %.2f s; resident %.1f GB'
          % (rep, dt, after[0] - before[0], after[2] - before[2], after[3] - before[3], rss))
    del h
