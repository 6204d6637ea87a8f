#!/usr/bin/env python3
"""Which V-CYCLE of a multigrid application needs the reference's forms on the finest
level (arithmetic='accurate')?  The solve with the fast forms in the first V-cycle(s)
of every application and the reference's in the later ones, against the oracle.

    python tools/history_by_cycle.py --configs square:3:6,square:5:8,square:6:9,lshape:5:8
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='square:3:6,square:5:8,square:6:9,lshape:5:8')
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'history_by_cycle.json'))
    args = ap.parse_args()
    import torch
    out = {}
    for spec in args.configs.split(','):
        problem, jt, js = spec.split(':')
        jt, js = int(jt), int(js)
        g = np.load(os.path.join(REPO, 'tests', 'golden', 'o1_pcg_%s_J%d_J%d.npz' % (problem, jt, js)))
        ref = np.asarray(g['hist'])
        rec = out.setdefault(spec, {})
        mg.GS_ALT_COPIES = True
        try:
            h = hm.HeatEquationMPI(J_space=js, J_time=jt, problem=problem, arithmetic='accurate')
        finally:
            mg.GS_ALT_COPIES = False
        for until, parts in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (2, 0)):
            for dev in (h.Kinv_x._dev, h.C_family._dev):
                dev.set_option('fast_until_cycle', until)
                dev.set_option('fast_parts', parts)
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
            name = 'fast forms in the first %d of 2 V-cycles' % until + (
                '' if not parts else ' + later ' + ' and '.join(
                    w for b, w in ((1, 'pre-smoothing'), (2, 'residual')) if parts & b))
            rec[name] = {'iterations': it, 'oracle_iterations': int(g['iters']), 'max_rel_dev': float(rel.max()),
                         'rel_dev_per_entry': [float(v) for v in rel], 'solve_s': solve}
            print('%-14s %-78s iters %2d/%2d  max %.1e  solve %.3f s' % (spec, name, it, int(g['iters']), rel.max(), solve),
                  flush=True)
        del h
        torch.cuda.empty_cache()
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(out, open(args.out, 'w'), indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
