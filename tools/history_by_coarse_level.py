#!/usr/bin/env python3
"""Which COARSE LEVELS own the gap between the batched preconditioner family and the
reference's one-hierarchy-per-wavelet-level (profiles/r06_history_attribution_J7_J10.json:
9.6e-11 against 1.5e-11 at config 5)?  The family combines two Galerkin chains per time
slice, ca (R A P) + cm (R M P); the reference forms the chain of the ASSEMBLED matrix
2^j M + alpha A (heateq_mpi.py:97-98, multigrid.py:142-145).  Both are sums of the same
products, rounded in different places: one ulp per entry.

This tool solves with family='reference' (one MultiGrid per wavelet level) and replaces, on
the levels l >= CUT of every member's chain (the finest level excepted: it is the
assembled matrix in both), the chain's matrix by the family's combination.  CUT = J: the
reference's chains everywhere; CUT = 0: the family's matrices everywhere (in the
reference's structure).  The history deviation as a function of CUT says which levels
must carry the reference's matrices for how much margin.

    python tools/history_by_coarse_level.py --J_time 7 --J_space 10
"""
import argparse
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source import multigrid as mg  # noqa: E402
from source.linalg import PCG  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--J_time', type=int, default=7)
    ap.add_argument('--J_space', type=int, default=10)
    ap.add_argument('--problem', default='square')
    ap.add_argument('--cuts', default='')
    ap.add_argument('--below', action='store_true',
                    help='replace the levels l < CUT instead (the small levels carry the family\'s matrices)')
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'history_by_coarse_level.json'))
    args = ap.parse_args()
    import torch
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'o1_pcg_%s_J%d_J%d.npz' % (args.problem, args.J_time,
                                                                                   args.J_space)))
    ref = np.asarray(g['hist'])
    alpha = 0.3
    state = {}

    def same(a, b):
        return a.shape == b.shape and a.nnz == b.nnz and np.array_equal(a.data, b.data)

    def hook(fine, mats):
        """mats: chain of `fine`, coarse to fine.  The chains of A_x and M_x themselves are
        kept for the combinations; the chains of 2^j M + alpha A get the family's matrices on
        the chosen levels."""
        A, M = state['A'], state['M']
        if same(fine, A):
            state['chain_a'] = list(mats)
            return mats
        if same(fine, M):
            state['chain_m'] = list(mats)
            return mats
        cm = (fine.diagonal()[0] - alpha * A.diagonal()[0]) / M.diagonal()[0]
        cm = 2.0**round(np.log2(cm))
        assert abs(sp.csr_matrix(fine - (cm * M + alpha * A))).max() == 0.0
        out = list(mats)
        J = len(mats) - 1
        for l in range(J):
            if (l < state['cut']) if args.below else (l >= state['cut']):
                combo = sp.csr_matrix(alpha * state['chain_a'][l] + cm * state['chain_m'][l])
                combo.sort_indices()
                out[l] = combo
        return out

    out = {}
    from source.assembly import space_matrices
    from source.multigrid import MeshHierarchy
    from source.problem import problem_helper
    mesh = problem_helper(args.problem, J_space=args.J_space, J_time=args.J_time)[0]
    M_x, A_x = space_matrices(mesh)
    state.update(A=sp.csr_matrix(A_x), M=sp.csr_matrix(M_x), cut=10**6)
    # the two pure chains, once (the hook records them)
    mg.CHAIN_HOOK = hook
    hier = MeshHierarchy(mesh)
    J = hier.J
    mg.MultiGrid(A_x, hier, smoothsteps=1, vcycles=1)
    mg.MultiGrid(M_x, hier, smoothsteps=1, vcycles=1)
    del hier
    torch.cuda.empty_cache()
    for cut in ([int(c) for c in args.cuts.split(',')] if args.cuts else list(range(J, -1, -1))):
        state['cut'] = cut
        mg.CHAIN_HOOK = hook
        try:
            h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time, problem=args.problem,
                                   family='reference')
        finally:
            mg.CHAIN_HOOK = None
        hist = []
        _, it = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
        hist = np.asarray(hist)
        n = min(len(hist), len(ref))
        rel = np.abs(hist[:n] / ref[:n] - 1.0)
        what = 'levels < %d' % cut if args.below else 'levels >= %d (of %d, finest excepted)' % (cut, J)
        out[str(cut)] = {'family_matrices_on': what, 'iterations': it, 'first_entry_rel_dev': float(rel[0]),
                         'max_rel_dev': float(rel.max())}
        print('family\'s matrices on %-40s iters %2d  first %.1e  max %.1e' % (what, it, rel[0], rel.max()),
              flush=True)
        del h
        torch.cuda.empty_cache()
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
