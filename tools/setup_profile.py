#!/usr/bin/env python3
"""Where the host time of the set-up goes: cProfile of the pieces of
HeatEquationMPI.__init__ run one after the other (the driver overlaps them)."""
import ctypes
import os as _os
import sys as _sys
# tools/setup_malloc_ab.sh: allocator settings made before anything starts a thread
if _os.environ.get('STK_EARLY_ARENA') == '1':
    ctypes.CDLL(None).mallopt(-8, 1)
if _os.environ.get('STK_HEAP') == '1':  # what bench.py and the drivers do
    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                      'spacetime-fullgrid-parallel_amd'))
    from source.host_malloc import keep_to_the_heap
    keep_to_the_heap()
import argparse
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import torch  # noqa: E402,F401
from source.assembly import space_load, space_matrices  # noqa: E402
from source.linop import EllMatrices  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--top', type=int, default=18)
ap.add_argument('--timeline', action='store_true', help='stages of HeatEquationMPI.__init__ as the driver runs them (threads and all), three times')
args = ap.parse_args()
torch.zeros(1, device='cuda')


if args.timeline:
    import heateq_mpi as hm
    if os.environ.get('STK_SWITCH_INTERVAL'):
        sys.setswitchinterval(float(os.environ['STK_SWITCH_INTERVAL']))
    for rep in range(3):
        t = time.time()
        h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time)
        torch.cuda.synchronize()
        print('set-up %d: %.2f s' % (rep, time.time() - t))
        last = 0.0
        for label, at in h.setup_timeline:
            print('   %-48s +%.2f s  (at %.2f s)' % (label, at - last, at))
            last = at
        for label, begin, end in sorted(getattr(h, 'setup_threads', []), key=lambda r: r[1]):
            print('      beside: %-28s %.2f -> %.2f s' % (label, begin, end))
        del h
    sys.exit(0)


def section(name, fn):
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    out = fn()
    pr.disable()
    print('==== %s: %.2f s' % (name, time.time() - t), flush=True)
    pstats.Stats(pr).sort_stats('tottime').print_stats(args.top)
    return out


mesh = section('problem_helper', lambda: problem_helper('square', J_space=args.J_space, J_time=args.J_time))
mesh_space, data = mesh[0], mesh[3]
M_x, A_x = section('space_matrices', lambda: space_matrices(mesh_space))
section('space_load', lambda: space_load(mesh_space, data['u0']))
hier = section('MeshHierarchy', lambda: MeshHierarchy(mesh_space))
section('EllMatrices', lambda: EllMatrices([M_x, A_x]))
section('MultiGrid(A_x)', lambda: MultiGrid(A_x, hier, smoothsteps=3, vcycles=2))
section('MultiGridFamily', lambda: MultiGridFamily(A_x, M_x, hier, ca=0.3, cms=[2**j for j in range(args.J_time + 1)],
                                                   smoothsteps=3, vcycles=2))
