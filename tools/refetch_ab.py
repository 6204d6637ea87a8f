#!/usr/bin/env python3
"""A/B of the launch and store options that decide how often the row-gather kernels
re-fetch their input: the last stage of S (kron_pack_kernel with an
input per term), I kron A_x (rows_ell_kernel<SPMM, 5>), and -- as wholes -- K^-1 and P,
whose V-cycles run the other SPMM instantiations.  Variants are sets of stk_set_tuning
keys; they are timed interleaved (HIP events), and every variant's outputs are compared
bit for bit with the first one's.  With ONE variant and few --reps the script is the
target of the counter passes (tools/pmc_passes.sh).

    python tools/refetch_ab.py --variants "base;nt:rows_nt_store=1;wg1:pack_multi_wg_per_cu=1"
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
import heateq_mpi as hm  # noqa: E402
from source import _lib  # noqa: E402

DEFAULTS = {'rows_nt_store': 0, 'rows_wg_per_cu': 0, 'rows_alternate': 1, 'pack_wg_per_cu': 0,
            'pack_flags': 3, 'pack_multi_wg_per_cu': 0, 'pack_multi_r': 0, 'pack_multi_lanes': 1,
            'terms_wg_per_cu': 0, 'terms_flags': 3, 'terms_r': 0, 'mg_coarse_static_fetch': 0}

ap = argparse.ArgumentParser()
ap.add_argument('--J_time', type=int, default=6)
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--problem', default='square')
ap.add_argument('--variants', default='base')
ap.add_argument('--ops', default='multi,ax,Kinv,P')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--rounds', type=int, default=3)
args = ap.parse_args()

variants = []
for spec in args.variants.split(';'):
    name, _, kv = spec.partition(':')
    keys = dict(DEFAULTS)
    for item in filter(None, kv.split(',')):
        k, v = item.split('=')
        assert k in DEFAULTS, k
        keys[k] = int(v)
    variants.append((name, keys))


def tune(keys):
    for k, v in keys.items():
        _lib.check(_lib.lib().stk_set_tuning(k.encode(), v))


h = hm.HeatEquationMPI(J_space=args.J_space, J_time=args.J_time, problem=args.problem)
S = h.S
n_loc, ld = h.rhs.n_loc, h.rhs.ld
torch.manual_seed(5)
x = torch.rand_like(h.rhs.buf)
v1, v2 = torch.rand_like(x), torch.rand_like(x)
for t in (x, v1, v2):
    t[:, n_loc:] = 0
packed = S.ell.packed_for(n_loc)
y = torch.empty_like(x)
A_x = h.CAC_j[0].linops[1]
from source.mpi_vector import KronVectorMPI  # noqa: E402
xv = KronVectorMPI(h.dofs_distr, x[:, :n_loc].t().contiguous().cpu().numpy())

ops = {
    'multi': lambda: packed.apply_multi([(None, 0, v1), (None, 1, v2), (S.tG, 0, x)], n_loc, ld, 0.0, y,
                                        steps=[None, None, (0, 1)]),
    'multi_nosteps': lambda: packed.apply_multi([(None, 0, v1), (None, 1, v2), (S.tG, 0, x)], n_loc, ld, 0.0, y),
    'ax': lambda: A_x.apply(x, out=y, n_loc=n_loc),
    'Kinv': lambda: h.Kinv_x.apply(x, n_loc=n_loc),
    'P': lambda: (h.P @ xv).buf,
    'S': lambda: (h.S @ xv).buf,
}
want = [o for o in args.ops.split(',') if o]


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


best = {}
ref = {}
for rnd in range(args.rounds):
    for name, keys in variants:
        tune(keys)
        for o in want:
            ms = timed(ops[o], args.reps)
            best[(name, o)] = min(best.get((name, o), 1e9), ms)
            if rnd == 0:
                out = ops[o]()
                out = (y if out is None else out).clone()
                torch.cuda.synchronize()
                if o not in ref:
                    ref[o] = out
                elif not torch.equal(out, ref[o]):
                    print('NOT bit-identical: %s %s' % (name, o))
tune(DEFAULTS)
print('J_time=%d J_space=%d %s, n_loc=%d; best of %d rounds of %d launches, ms' % (
    args.J_time, args.J_space, args.problem, n_loc, args.rounds, args.reps))
print('%-34s' % 'variant' + ''.join('%10s' % o for o in want))
for name, keys in variants:
    diff = ','.join('%s=%d' % (k, v) for k, v in keys.items() if v != DEFAULTS[k])
    print('%-34s' % ('%s %s' % (name, diff))[:34] + ''.join('%10.3f' % best[(name, o)] for o in want))
