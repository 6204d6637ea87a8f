#!/usr/bin/env python3
"""Which rounding owns the gap between the GPU solve's r.Pr history and the CPU
oracle's?  Solves the BASELINE configurations with each of this build's
regroupings switched off in turn and compares every history with the oracle
fixture (tests/golden/o1_pcg_*.npz):

  fast               arithmetic='fast': fused Schur complement, batched multigrid family,
                     diagonal-free Gauss-Seidel rows, restricted residual (R A) u - R f
  schur=reference    the five-term sum of reference heateq_mpi.py:166-181
  gs=full rows       u_i += (f_i - row_i u) / a_ii (multigrid.py:89-97)
  family=reference   one hierarchy per wavelet level from the assembled
                     2^j M + alpha A (heateq_mpi.py:147-153)
  restrict=R(Au-f)   the restricted residual as the reference forms it
                     (multigrid.py:174-175) instead of (R A) u - R f
  arithmetic=accurate    the library's default: restrict=R(Au-f) + gs=full rows
  arithmetic=reference   all four

    python tools/history_attribution.py --configs square:5:8,square:6:9,lshape:5:8
Writes gpurun_out/history_attribution.json (copy it to profiles/).
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

# (name, HeatEquationMPI keywords, diagonal-free Gauss-Seidel rows, tuning keys)
VARIANTS = [
    ('fast', {}, True, {}),
    ('schur=reference', {'schur': 'reference'}, True, {}),
    ('gs=full rows', {}, False, {}),
    ('restrict=R(Au-f)', {}, True, {'mg_fuse_restrict': 0}),
    ('restrict=R(Au-f) + gs=full rows', {}, False, {'mg_fuse_restrict': 0}),
    ('family=reference', {'family': 'reference'}, True, {}),
    ('family=reference + restrict=R(Au-f)', {'family': 'reference'}, True, {'mg_fuse_restrict': 0}),
    ('family=reference + restrict=R(Au-f) + gs=full rows', {'family': 'reference'}, False,
     {'mg_fuse_restrict': 0}),
    ('arithmetic=accurate', {'arithmetic': 'accurate'}, True, {}),
    ('arithmetic=reference', {'arithmetic': 'reference'}, True, {}),
    # round 6 (config 5's margin): which of the two regroupings the default still makes owns
    # what is left -- the regrouped S or the family's ca RAP + cm RMP coarse matrices
    ('accurate + schur=reference', {'arithmetic': 'accurate', 'schur': 'reference'}, True, {}),
    ('accurate + family=reference', {'arithmetic': 'accurate', 'family': 'reference'}, True, {}),
    ('accurate + gs=full rows on every level', {'arithmetic': 'accurate', 'ACCURATE': {'gs_rows': 'full'}}, True, {}),
    ('accurate + R(Au-f) on every level', {'arithmetic': 'accurate',
                                           'ACCURATE': {'fuse_restrict_below_finest': False}}, True, {}),
    ('accurate + every V-cycle in the reference forms', {'arithmetic': 'accurate',
                                                         'ACCURATE': {'fast_leading_cycles': False}}, True, {}),
    ('accurate + pre-smoothing of the last V-cycle in the reference form',
     {'arithmetic': 'accurate', 'ACCURATE': {'fast_parts': 0}}, True, {}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='square:3:6,square:5:8,square:6:9,lshape:5:8')
    ap.add_argument('--variants', default='')
    ap.add_argument('--accurate-knobs', action='store_true',
                    help="the default arithmetic, the reference arithmetic and the default with one more "
                         "of the reference's forms each (round 6: who owns config 5's 9.6e-11)")
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'history_attribution.json'))
    args = ap.parse_args()
    import torch
    want = set(filter(None, args.variants.split(',')))
    if args.accurate_knobs:
        want = {v[0] for v in VARIANTS if v[0].startswith('accurate') or v[0] in ('arithmetic=accurate',
                                                                                  'arithmetic=reference')}
    out = {}
    for spec in args.configs.split(','):
        problem, jt, js = spec.split(':')
        jt, js = int(jt), int(js)
        g = np.load(os.path.join(REPO, 'tests', 'golden', 'o1_pcg_%s_J%d_J%d.npz' % (problem, jt, js)))
        ref = np.asarray(g['hist'])
        rec = out.setdefault(spec, {})
        from source import _lib
        for name, kw, diag_free, tune in VARIANTS:
            if want and name not in want:
                continue
            for key, value in tune.items():
                _lib.check(_lib.lib().stk_set_tuning(key.encode(), value))
            mg.GS_DIAG_FREE = diag_free
            t0 = time.time()
            kw = dict(kw)
            default = dict(hm.HeatEquationMPI.ACCURATE)
            hm.HeatEquationMPI.ACCURATE = dict(default, **kw.pop('ACCURATE', {}))
            try:
                h = hm.HeatEquationMPI(J_space=js, J_time=jt, problem=problem, **dict({'arithmetic': 'fast'}, **kw))
            finally:
                hm.HeatEquationMPI.ACCURATE = default
            mg.GS_DIAG_FREE = True
            setup = time.time() - t0
            hist = []
            torch.cuda.synchronize()
            t0 = time.time()
            _, it = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
            torch.cuda.synchronize()
            solve = time.time() - t0
            for key in tune:
                _lib.check(_lib.lib().stk_set_tuning(key.encode(), 1))
            hist = np.asarray(hist)
            n = min(len(hist), len(ref))
            rel = np.abs(hist[:n] / ref[:n] - 1.0)
            rec[name] = {'iterations': it, 'oracle_iterations': int(g['iters']),
                         'first_entry_rel_dev': float(rel[0]), 'max_rel_dev': float(rel.max()),
                         'rel_dev_per_entry': [float(v) for v in rel],
                         'max_dev_relative_to_initial': float(np.abs(hist[:n] - ref[:n]).max() / ref[0]),
                         'setup_s': setup, 'solve_s': solve}
            print('%-14s %-34s iters %2d/%2d  first %.1e  max %.1e  (setup %.1f s, solve %.2f s)' % (
                spec, name, it, int(g['iters']), rel[0], rel.max(), setup, solve), flush=True)
            del h
            torch.cuda.empty_cache()
            os.makedirs(os.path.dirname(args.out), exist_ok=True)
            json.dump(out, open(args.out, 'w'), indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
