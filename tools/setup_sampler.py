#!/usr/bin/env python3
"""Where the WALL time of HeatEquationMPI.__init__ goes, thread by thread: a sampler thread
reads sys._current_frames() every 2 ms during three set-ups and counts, per thread, the
innermost frame inside this repository (and the NumPy / SciPy / torch call it sits in).
cProfile sees one thread and charges lock waits to it; this sees all of them."""
import collections
import os
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source.host_malloc import keep_to_the_heap  # noqa: E402
keep_to_the_heap()  # the allocator policy of the drivers (STK_KEEP_MALLOC=1: untouched)
import torch  # noqa: E402
import heateq_mpi as hm  # noqa: E402

J_space = int(sys.argv[1]) if len(sys.argv) > 1 else 9
J_time = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.zeros(1, device='cuda')
hm.HeatEquationMPI(J_space=J_space, J_time=J_time)  # warm-up
counts = collections.Counter()
leaf = collections.Counter()
stop = False


def sampler(me):
    while not stop:
        for tid, frame in sys._current_frames().items():
            if tid == me:
                continue
            f, inner = frame, None
            while f is not None:
                fn = f.f_code.co_filename
                if REPO in fn and 'setup_sampler' not in fn:
                    inner = '%s:%s' % (os.path.basename(fn), f.f_code.co_name)
                    break
                f = f.f_back
            if inner is None:
                continue
            counts[inner] += 1
            top = frame.f_code
            leaf[(inner, '%s:%s:%d' % (os.path.basename(top.co_filename), top.co_name, frame.f_lineno))] += 1
        time.sleep(0.002)


t = threading.Thread(target=sampler, args=(None,), daemon=True)
t = threading.Thread(target=lambda: sampler(threading.get_ident()), daemon=True)
t.start()
t0 = time.time()
for _ in range(3):
    h = hm.HeatEquationMPI(J_space=J_space, J_time=J_time)
    torch.cuda.synchronize()
    del h
wall = (time.time() - t0) / 3
stop = True
t.join()
total = sum(counts.values())
print('set-up %.2f s on average; %d samples (2 ms apart, all threads)' % (wall, total))
print('-- innermost repository frame, share of all thread-samples')
for name, c in counts.most_common(25):
    print('%6.1f %%  %s' % (100.0 * c / total, name))
print('-- (repository frame, innermost line)')
for (name, where), c in leaf.most_common(40):
    print('%6.1f %%  %-40s %s' % (100.0 * c / total, name, where))
