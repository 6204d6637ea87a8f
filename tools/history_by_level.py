#!/usr/bin/env python3
"""On WHICH LEVELS do the two regroupings of arithmetic='fast' (diagonal-free
Gauss-Seidel rows, restricted residual as (R A) u - R f) move the solve's r.Pr
history away from the CPU oracle's?  The solve with the fast forms on some levels
and the reference's on the others, against the oracle fixtures.

    python tools/history_by_level.py --configs square:5:8,square:6:9,lshape:5:8
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source import multigrid as mg  # noqa: E402
from source.linalg import PCG  # noqa: E402

BIG = 1 << 30
# name, levels with diagonal-free rows (level, finest) -> bool, fused restriction on levels [lo(J), hi(J)]
VARIANTS = [
    ('fast on every level', lambda j, J: True, lambda J: (0, BIG)),
    ('reference forms on every level (= accurate)', lambda j, J: False, lambda J: (BIG, BIG)),
    ('fast on the finest level only', lambda j, J: j == J, lambda J: (J, BIG)),
    ('fast on the two finest levels', lambda j, J: j >= J - 1, lambda J: (J - 1, BIG)),
    ('fast below the finest level only', lambda j, J: j < J, lambda J: (0, J - 1)),
    ('rows: fast on the finest only; residual: reference everywhere', lambda j, J: j == J, lambda J: (BIG, BIG)),
    ('rows: reference everywhere; residual: fast on the finest only', lambda j, J: False, lambda J: (J, BIG)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='square:5:8,square:6:9,lshape:5:8')
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'history_by_level.json'))
    args = ap.parse_args()
    import torch
    out = {}
    for spec in args.configs.split(','):
        problem, jt, js = spec.split(':')
        jt, js = int(jt), int(js)
        g = np.load(os.path.join(REPO, 'tests', 'golden', 'o1_pcg_%s_J%d_J%d.npz' % (problem, jt, js)))
        ref = np.asarray(g['hist'])
        rec = out.setdefault(spec, {})
        for name, rows, fused in VARIANTS:
            mg.GS_DIAG_FREE_LEVELS = rows
            try:
                h = hm.HeatEquationMPI(J_space=js, J_time=jt, problem=problem, arithmetic='fast')
            finally:
                mg.GS_DIAG_FREE_LEVELS = None
            lo, hi = fused(h.hierarchy.J)
            for dev in (h.Kinv_x._dev, h.C_family._dev):
                dev.set_option('fuse_restrict_min_level', lo)
                dev.set_option('fuse_restrict_max_level', hi)
            hist = []
            PCG(h.WT_S_W, h.P, h.rhs, kmax=4)
            torch.cuda.synchronize()
            t0 = time.time()
            _, it = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
            torch.cuda.synchronize()
            solve = time.time() - t0
            hist = np.asarray(hist)
            n = min(len(hist), len(ref))
            rel = np.abs(hist[:n] / ref[:n] - 1.0)
            rec[name] = {'iterations': it, 'oracle_iterations': int(g['iters']), 'max_rel_dev': float(rel.max()),
                         'rel_dev_per_entry': [float(v) for v in rel], 'solve_s': solve}
            print('%-14s %-62s iters %2d/%2d  max %.1e  solve %.3f s' % (spec, name, it, int(g['iters']), rel.max(), solve),
                  flush=True)
            del h
            torch.cuda.empty_cache()
            os.makedirs(os.path.dirname(args.out), exist_ok=True)
            json.dump(out, open(args.out, 'w'), indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
