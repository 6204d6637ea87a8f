#!/usr/bin/env python3
"""Interleaved A/B timing of the metric's Kronecker apply under tuning variants
(one process, rounds alternate the variants, median of rounds), plus the
per-segment cycle shares of the stamped diagnostic build.

    python tools/kron_ab.py --variants "plain;pack;pack,pack_flags=3;pack,pack_block=256"
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib  # noqa: E402
from source.assembly import space_matrices  # noqa: E402
from source.linop import EllMatrices  # noqa: E402
from source.problem import problem_helper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--problem', default='square')
ap.add_argument('--n_loc', type=int, default=65)
ap.add_argument('--ghosts', type=int, default=0)
ap.add_argument('--ld', type=int, default=0, help='row stride in doubles (0: n_loc rounded up to even)')
ap.add_argument('--rounds', type=int, default=9)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--variants', default='plain;pack;pack,pack_flags=1;pack,pack_flags=2;pack,pack_flags=3')
ap.add_argument('--phases', action='store_true')
ap.add_argument('--rows-per-tile', default='')
ap.add_argument('--order', default='patch', help='patch (assembly hint) or index (memory order)')
args = ap.parse_args()
if args.rows_per_tile:
    os.environ['STK_ROWS_PER_TILE'] = args.rows_per_tile
mesh = problem_helper(args.problem, J_space=args.J_space, J_time=2)[0]
M_x, A_x = space_matrices(mesh)
M = M_x.shape[0]
if args.order == 'index':
    M_x.stk_row_order = A_x.stk_row_order = np.arange(M, dtype=np.int32)
ell = EllMatrices([M_x, A_x], [M_x])
n_loc = args.n_loc
ld = args.ld or (n_loc + (n_loc & 1))
rng = np.random.RandomState(0)
x = torch.rand((M, ld), dtype=torch.float64, device='cuda')
x[:, n_loc:] = 0
y = torch.empty_like(x)
tri = [_lib.to_dev(rng.rand(3, n_loc)) for _ in range(2)]
g = torch.rand((2, M), dtype=torch.float64, device='cuda') if args.ghosts else None
gh = None
if g is not None:
    gh = torch.empty((M, 2), dtype=torch.float64, device='cuda')
    _lib.check(_lib.lib().stk_interleave_ghosts(_lib.stream(), M, _lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(gh)))
lo, hi = (g[0], g[1]) if g is not None else (None, None)
nbytes = 16 * n_loc * M + 8 * (2 if g is not None else 0) * M + 12 * (M_x.nnz + A_x.nnz) + 8 * (M + 1)
KEYS = {'pack_flags': 3, 'pack_block': 512, 'pack_wg_per_cu': 0, 'ell_wg_per_cu': 0}


def run(variant):
    parts = variant.split(',')
    for k, v in KEYS.items():
        _lib.check(_lib.lib().stk_set_tuning(k.encode(), v))
    for kv in parts[1:]:
        k, v = kv.split('=')
        _lib.check(_lib.lib().stk_set_tuning(k.encode(), int(v)))
    if parts[0] == 'plain':
        return lambda: ell.apply([(tri[0], 0, x, lo, hi), (tri[1], 1, x, lo, hi)], n_loc, ld, 0.0, y)
    form = {'pack': ell.packed, 'pack1': forms[1], 'pack2': forms[2]}[parts[0]]
    return lambda: form.apply([(tri[0], 0), (tri[1], 1)], x, gh, n_loc, ld, 0.0, y)


forms = {rp: ell.packed_variant(rp) for rp in (1, 2)}
print('packed forms: default %d row(s) per unit' % ell.packed.rows_per_unit, flush=True)
y1 = torch.full_like(x, 3.0)
if forms[1].ok and forms[1].rows_per_unit == 1:
    forms[1].apply([(tri[0], 0), (tri[1], 1)], x, gh, n_loc, ld, 0.0, y1)
else:  # values without a dictionary: the one-row form is the plain one
    print('   no dictionary for these values: one-row form = plain sliced ELL', flush=True)
    ell.apply([(tri[0], 0, x, lo, hi), (tri[1], 1, x, lo, hi)], n_loc, ld, 0.0, y1)
for rp in (2,):
    f = forms[rp]
    if f.rows_per_unit != rp:
        print('   %d rows per unit: not available for these matrices' % rp)
        continue
    y2 = torch.full_like(x, 5.0)
    f.apply([(tri[0], 0), (tri[1], 1)], x, gh, n_loc, ld, 0.0, y2)
    torch.cuda.synchronize()
    same = torch.equal(y1[:, :n_loc], y2[:, :n_loc])  # (a column range of a wider slab leaves the other columns alone)
    print('   %d rows per unit: %d units for %d rows, K=%d, %s; bit-identical with single rows: %s'
          % (rp, f.n_units, M, f.K, 'explicit values' if f.explicit else '%d codes' % f.n_codes, same), flush=True)
    assert same


variants = args.variants.split(';')
times = {v: [] for v in variants}
for rnd in range(args.rounds):
    for v in variants:
        fn = run(v)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / args.reps)
for v in variants:
    t = np.array(times[v])
    med = float(np.median(t))
    print('%-40s median %.4f ms  min %.4f  max %.4f   %.0f GB/s algorithmic = %.1f%% of 8 TB/s'
          % (v, med, t.min(), t.max(), nbytes / med / 1e6, nbytes / med / 1e6 / 80))

if args.phases and g is None:
    for k, v in KEYS.items():
        _lib.check(_lib.lib().stk_set_tuning(k.encode(), v))
    buf = torch.zeros((4096 * 8, 4), dtype=torch.int64, device='cuda')
    _lib.check(_lib.lib().stk_kron_pack_set_diag(buf.data_ptr()))
    fn = run('pack1')
    for _ in range(3):
        buf.zero_()
        fn()
    torch.cuda.synchronize()
    first = buf.cpu().numpy().astype(np.float64)
    # ablations of the stamped build (timing only)
    for flags, what in ((0, 'stamped build as is'), (16, 'all slots gather the own column'),
                        (32, 'one gather per lane'), (48, 'one gather, own column')):
        _lib.check(_lib.lib().stk_set_tuning(b'pack_flags', flags))
        _lib.check(_lib.lib().stk_kron_pack_set_diag(buf.data_ptr()))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print('   ablation %-36s %.4f ms' % (what, e0.elapsed_time(e1) / args.reps))
    _lib.check(_lib.lib().stk_set_tuning(b'pack_flags', 0))
    _lib.check(_lib.lib().stk_kron_pack_set_diag(None))
    d = first
    d = d[d.sum(axis=1) > 0]
    tot = d.sum(axis=1)
    names = ['publish + barrier 1', 'gathers + space factors', 'exchange + barrier 2', 'time stencil + store']
    print('stamped build: %d waves, %.0f cycles per wave in the loop (median), shares:' % (len(d), np.median(tot)))
    for q, nme in enumerate(names):
        print('   %-26s %5.1f %%   (%.0f cycles per wave)' % (nme, 100 * d[:, q].sum() / tot.sum(), np.median(d[:, q])))
