#!/usr/bin/env python3
"""Times the Kronecker kernel in the slab shapes the ranks of a 2/4/8-GPU run
see at J_time=6, J_space=9 (n_loc = 33/17/9 with one or two ghost rows), on ONE
GPU: the ghost rows are filled with random data instead of being received.
Gives the compute part of a multi-GPU step without a multi-GPU node."""
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
from source.mesh import construct_2d_square_mesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--J_space', type=int, default=9)
ap.add_argument('--shapes', default='65:0:0,33:0:1,32:1:0,17:1:1,16:1:1,9:1:1,8:1:1')
ap.add_argument('--tune', default='')  # key=value,...
ap.add_argument('--flags', default='0,1,2,3')  # pack_flags values to time
args = ap.parse_args()
for kv in filter(None, args.tune.split(',')):
    k, v = kv.split('=')
    _lib.check(_lib.lib().stk_set_tuning(k.encode(), int(v)))

mesh, _ = construct_2d_square_mesh(args.J_space)
M_x, A_x = space_matrices(mesh)
M = M_x.shape[0]
ell = EllMatrices([M_x, A_x], [M_x])
rng = np.random.RandomState(0)
for shape in args.shapes.split(','):
    f = [int(v) for v in shape.split(':')]
    n_loc, lo, hi = f[:3]
    ld = f[3] if len(f) > 3 else n_loc + (n_loc & 1)  # optional 4th field: row stride
    x = torch.rand((M, ld), dtype=torch.float64, device='cuda')
    x[:, n_loc:] = 0
    y = torch.empty_like(x)
    g = torch.rand((2, M), dtype=torch.float64, device='cuda')
    tri = [_lib.to_dev(rng.rand(3, n_loc)) for _ in range(2)]
    specs = [(tri[0], 0, x, g[0] if lo else None, g[1] if hi else None),
             (tri[1], 1, x, g[0] if lo else None, g[1] if hi else None)]
    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    ms = timed(lambda: ell.apply(specs, n_loc, ld, 0.0, y))
    ms_local = timed(lambda: ell.apply_local(specs, n_loc, ld, 0.0, y))
    ms_ghost = timed(lambda: ell.apply_ghost(specs, n_loc, ld, y))
    nbytes = 16 * n_loc * M + 8 * (lo + hi) * M + 12 * (M_x.nnz + A_x.nnz) + 8 * (M + 1)
    print('n_loc=%2d ghosts=%d%d  plain: %.3f ms (local %.3f + ghost rows %.3f)  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)'
          % (n_loc, lo, hi, ms, ms_local, ms_ghost, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
    # packed form: matrix stream 4 bytes per slot, ghost steps fused
    gh = None
    if lo or hi:
        gh = torch.empty((M, 2), dtype=torch.float64, device='cuda')
        _lib.check(_lib.lib().stk_interleave_ghosts(
            _lib.stream(), M, _lib.ptr(g[0] if lo else None), _lib.ptr(g[1] if hi else None), _lib.ptr(gh)))
    y2 = torch.empty_like(x)
    pspecs = [(tri[0], 0), (tri[1], 1)]
    for flags in [int(v) for v in args.flags.split(',')]:
        _lib.check(_lib.lib().stk_set_tuning(b'pack_flags', flags))
        one = ell.packed_variant(1)
        ms1 = timed(lambda: one.apply(pspecs, x, gh, n_loc, ld, 0.0, y2))
        msp = timed(lambda: ell.packed.apply(pspecs, x, gh, n_loc, ld, 0.0, y2))
        ell.apply(specs, n_loc, ld, 0.0, y)
        err = float((y2 - y).abs().max())
        print('                     packed, one row per slot row: %.3f ms' % ms1)
        print('                     packed flags=%d: %.3f ms  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)  max|diff to plain| %.1e  -> x%d ranks: %.2f TB/s aggregate'
              % (flags, msp, nbytes / msp / 1e6, nbytes / msp / 1e6 / 80, err, round(65 / n_loc),
                 round(65 / n_loc) * nbytes / msp / 1e9))
    _lib.check(_lib.lib().stk_set_tuning(b'pack_flags', 3))
    if lo or hi:
        # the WHOLE step of a rank but for the wire time: what the exchange sends is
        # packed out of the slab (stk_halo_pack) and what arrives enters the apply
        lib = _lib.lib()
        pk = ell.packed_for(n_loc)
        send = torch.empty((2, M), dtype=torch.float64, device='cuda')
        st = _lib.stream()

        def pack():
            _lib.check(lib.stk_halo_pack(st, M, n_loc, ld, _lib.ptr(x), _lib.ptr(send[0]) if lo else None, 1,
                                         _lib.ptr(send[1]) if hi else None, 1))

        def one_pass():  # wait for the halo, interleave it, one pass with ghost lanes
            pack()
            _lib.check(lib.stk_interleave_ghosts(st, M, _lib.ptr(g[0] if lo else None),
                                                 _lib.ptr(g[1] if hi else None), _lib.ptr(gh)))
            pk.apply(pspecs, x, gh, n_loc, ld, 0.0, y2)

        def overlapped():  # pass without ghost steps (hides the wire), then their share
            pack()
            pk.apply(pspecs, x, None, n_loc, ld, 0.0, y2)
            pk.apply_ghost(pspecs, x, g[0] if lo else None, g[1] if hi else None, n_loc, ld, y2)

        ms_pack = timed(pack)
        ms_a, ms_b = timed(one_pass), timed(overlapped)
        ms_main = timed(lambda: pk.apply(pspecs, x, None, n_loc, ld, 0.0, y2))
        ms_go = timed(lambda: pk.apply_ghost(pspecs, x, g[0] if lo else None, g[1] if hi else None, n_loc, ld, y2))
        # round 6: the boundary steps from compact records (left by the pack) and interleaved received rows
        rec = torch.empty((M, 4), dtype=torch.float64, device='cuda')

        def pack_records():
            _lib.check(lib.stk_halo_pack_records(st, M, n_loc, ld, _lib.ptr(x), _lib.ptr(send[0]) if lo else None, 1,
                                                 _lib.ptr(send[1]) if hi else None, 1, _lib.ptr(rec)))

        def interleave():
            _lib.check(lib.stk_interleave_ghosts(st, M, _lib.ptr(g[0] if lo else None),
                                                 _lib.ptr(g[1] if hi else None), _lib.ptr(gh)))

        pack_records()
        interleave()
        y3 = torch.empty_like(y2)
        pk.apply(pspecs, x, None, n_loc, ld, 0.0, y3)
        pk.apply_boundary(pspecs, rec, gh, lo, hi, n_loc, ld, y3)
        same = torch.equal(y3, y2)
        ms_pr, ms_il = timed(pack_records), timed(interleave)
        ms_rec = timed(lambda: pk.apply_boundary(pspecs, rec, gh, lo, hi, n_loc, ld, y3))
        print('                     boundary steps from compact records: pack with records %.3f ms; interleave %.3f ms; '
              'boundary kernel %.3f ms (slab gathers: %.3f ms)  %s' % (ms_pr, ms_il, ms_rec, ms_go,
                                                                        'bit-identical' if same else 'DIFFERENT'))
        overlapped()
        ell.apply(specs, n_loc, ld, 0.0, y)
        err = float((y2 - y).abs().max() / y.abs().max())
        print('                     step without the wire: pack %.3f ms; pack + interleave + one pass with ghost lanes %.3f ms; '
              'pack + pass without ghosts (%.3f, runs beside the exchange) + ghost share (%.3f) = %.3f ms, of which %.3f ms '
              'after the halo has arrived  (rel. diff to plain %.1e)' % (ms_pack, ms_a, ms_main, ms_go, ms_b,
                                                                        ms_go, err))
    _lib.check(_lib.lib().stk_set_tuning(b'pack_flags', 3))
