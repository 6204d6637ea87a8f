#!/usr/bin/env python3
"""cProfile of the multigrid plan construction with its threads run inline, so that
the profile sees the work of every level (tools/setup_profile.py only sees the
parent waiting).  Sequential time, not the set-up's wall time."""
import argparse
import cProfile
import concurrent.futures
import os
import pstats
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import torch  # noqa: E402,F401
from source.assembly import space_matrices  # noqa: E402
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--top', type=int, default=30)
ap.add_argument('--sort', default='cumulative')
args = ap.parse_args()
if torch.cuda.is_available():
    torch.zeros(1, device='cuda')


class InlineFuture:
    def __init__(self, fn, *a, **k):
        self._r = fn(*a, **k)

    def result(self):
        return self._r


class InlinePool:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def submit(self, fn, *a, **k):
        return InlineFuture(fn, *a, **k)


class InlineThread:
    def __init__(self, target=None, args=(), kwargs=None):
        self._t, self._a, self._k = target, args, kwargs or {}

    def start(self):
        self._t(*self._a, **self._k)

    def join(self):
        pass


concurrent.futures.ThreadPoolExecutor = InlinePool
threading.Thread = InlineThread
mesh = problem_helper('square', J_space=args.J_space, J_time=args.J_time)[0]
M_x, A_x = space_matrices(mesh)
hier = MeshHierarchy(mesh)
for name, fn in (('MultiGrid(A_x)', lambda: MultiGrid(A_x, hier, smoothsteps=3, vcycles=2)),
                 ('MultiGridFamily', lambda: MultiGridFamily(A_x, M_x, hier, ca=0.3,
                                                             cms=[2**j for j in range(args.J_time + 1)],
                                                             smoothsteps=3, vcycles=2))):
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    fn()
    pr.disable()
    print('==== %s inline: %.2f s' % (name, time.time() - t), flush=True)
    pstats.Stats(pr).sort_stats(args.sort).print_stats(args.top)
