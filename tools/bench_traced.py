#!/usr/bin/env python3
"""bench.py with a stack dump of every thread each 60 s into gpurun_out/bench_trace_rank<R>.txt
(faulthandler): where a multi-rank run that takes long -- or hangs -- is at.  Arguments as bench.py's;
start it the way bench.py is started (torch.distributed.run for several ranks)."""
import faulthandler
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(REPO, 'gpurun_out')
os.makedirs(out, exist_ok=True)
trace = open(os.path.join(out, 'bench_trace_rank%s.txt' % os.environ.get('RANK', '0')), 'w')
faulthandler.dump_traceback_later(60, repeat=True, file=trace)
sys.argv[0] = os.path.join(REPO, 'bench.py')
runpy.run_path(sys.argv[0], run_name='__main__')
