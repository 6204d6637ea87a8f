#!/usr/bin/env python3
"""PCG trajectories of the CPU ORACLE at the sizes of the BASELINE configs
(not from the reference: its pure-Python smoother would need days at these
sizes, and NGSolve is absent).  The oracle is pinned to the reference by
tests/test_oracle_golden.py; these fixtures extend the GPU parity check of the
whole solve -- iteration count, r.Pr history, sampled solution, as the
reference's integration test compares them (heateq_mpi_test.py:138-189) -- to
full size.

    python tests/golden/make_oracle_vectors.py --J_time 5 --J_space 8
    python tests/golden/make_oracle_vectors.py --J_time 6 --J_space 9 --threads 7
    python tests/golden/make_oracle_vectors.py --J_time 5 --J_space 8 --problem lshape

Time slices are independent in every space operator, so --threads only cuts
batches of slices into chunks; the numbers do not depend on it.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

from oracle import multigrid as omg  # noqa: E402
from oracle.heat import HeatEquationOracle  # noqa: E402
from oracle.krylov import pcg  # noqa: E402
from source.assembly import (prolongation_matrices, space_load,  # noqa: E402
                             space_matrices, time_matrices)
from source.problem import problem_helper  # noqa: E402


def fixture_name(problem, J_time, J_space, precond='multigrid'):
    return 'o1_pcg_%s_J%d_J%d%s.npz' % (problem, J_time, J_space, '' if precond == 'multigrid' else '_' + precond)


def sample_strides(N, M):
    """Sub-sampling of an (N, M) array kept in the fixture: about 8 time rows
    by about 200 space dofs (always including row 0 and the last row)."""
    return max(1, (N - 1) // 8), max(1, M // 199)


def build_oracle(problem, J_time, J_space, precond='multigrid'):
    mesh, _, tmesh, data, _ = problem_helper(problem, J_space, J_time)
    A_t, L_t, M_t, G_t, u0_t = time_matrices(tmesh)
    M_x, A_x = space_matrices(mesh, scipy_path=True)  # the generator does not load libstk
    mats = dict(A_t=A_t, L_t=L_t, M_t=M_t, G_t=G_t, M_x=M_x, A_x=A_x,
                P_mats=prolongation_matrices(mesh), u0_t=u0_t,
                u0_x=space_load(mesh, data['u0'], numpy_path=True))  # the form the fixtures were made with
    return HeatEquationOracle(mats, J_time, precond=precond)


def lean_operators(o, chunk=8):
    """S and W^T S W of the oracle evaluated in chunks of time rows: (S x)[t] only
    needs the rows t-1, t, t+1 of the time factors, so the (M, N) temporaries of
    HeatEquationOracle.S -- several copies of the vector -- shrink to `chunk`
    rows.  Same arithmetic per entry (the space operators act slice by slice).
    For the largest configuration (J_time=7, J_space=10: 4.3 GB per vector)."""
    import scipy.sparse as sp
    from oracle import kron
    terms = [(sp.csr_matrix(T), ops) for T, ops in o.S_terms()]

    def S(X):
        out = np.zeros_like(X)
        for a in range(0, o.N, chunk):
            b = min(a + chunk, o.N)
            for T, ops in terms:
                Z = T[a:b] @ X  # rows a..b of (T kron I) x
                out[a:b] += kron.composite_space(ops, Z.T).T
        return out

    return S, (lambda X: o.WT(S(o.W(X))))


def lean_pcg(T, P, b, kmax, keep_at=(), sample=None, progress=None):
    """oracle.krylov.pcg with every vector update done in place, row block by row
    block (identical per entry; no full-size temporaries).  `keep_at`: iteration
    numbers after which norm and sample of the iterate are kept (the head of a
    long trajectory stays comparable on its own); `progress`: a JSON file that
    holds the history so far (a run of hours that dies keeps what it had)."""
    import json
    from oracle.krylov import _dot
    N = b.shape[0]
    kept = {}

    def axpy(y, a, x):  # y += a * x
        for t in range(N):
            y[t] += a * x[t]

    w = np.zeros_like(b)
    r = b.copy()  # w0 = 0: r = b - T(0) = b
    p = P(r)
    abs_r = _dot(r, p)
    hist = [abs_r]
    for k in range(1, kmax):
        t = T(p)
        alpha = abs_r / _dot(p, t)
        axpy(w, alpha, p)
        axpy(r, -alpha, t)
        del t
        z = P(r)
        abs_r_old, abs_r = abs_r, _dot(r, z)
        hist.append(abs_r)
        print('  iteration %d: r.Pr = %.6e' % (k, abs_r), flush=True)
        if k in keep_at:
            kept[k] = (np.linalg.norm(w), sample(w).copy())
        if progress:
            with open(progress, 'w') as f:
                json.dump(dict(hist=[float.hex(float(v)) for v in hist]), f)
        if abs_r < 1e-12:
            break
        beta = abs_r / abs_r_old
        for row in range(N):
            p[row] *= beta
            p[row] += z[row]
        del z
    return w, len(hist) - 1, hist, kept


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--J_time', type=int, default=5)
    ap.add_argument('--J_space', type=int, default=6)
    ap.add_argument('--problem', default='square')
    ap.add_argument('--threads', type=int, default=1)
    ap.add_argument('--kmax', type=int, default=100000)
    ap.add_argument('--precond', default='multigrid',
                    help="'direct': K and the blocks of P as sparse direct inverses (reference "
                         "heateq_mpi.py:154-157, heateq_mpi_test.py:66-135); fixture ..._direct.npz")
    ap.add_argument('--no-ops', action='store_true',
                    help='skip the S(X), P(X) samples (memory at the largest sizes)')
    ap.add_argument('--ops-only', action='store_true',
                    help='keep the trajectory of the existing fixture and recompute only the '
                         'S / P / W samples of the bench vector (after a change of that vector)')
    ap.add_argument('--lean', action='store_true',
                    help='chunked S and in-place PCG (the largest configuration, with --kmax 3: '
                         'the first entries of the history only)')
    args = ap.parse_args()
    omg.THREADS = args.threads
    o = build_oracle(args.problem, args.J_time, args.J_space, args.precond)
    t0 = time.time()

    def cb(w, r, k):
        print('  iteration %d  (%.0f s)' % (k, time.time() - t0), flush=True)

    path = os.path.join(HERE, fixture_name(args.problem, args.J_time, args.J_space, args.precond))
    st, sx = sample_strides(o.N, o.M)
    if args.ops_only:
        have = np.load(path)
        out = {k: have[k] for k in have.files}
        assert (int(out['J_time']), int(out['J_space']), str(out['problem'])) == (
            args.J_time, args.J_space, args.problem)
        assert tuple(int(v) for v in out['sample_strides']) == (st, sx)
        if args.lean:
            o.S = lean_operators(o)[0]
    else:
        if args.lean:
            S_lean, T_lean = lean_operators(o)
            o.S = S_lean
            w, iters, hist, kept = lean_pcg(
                T_lean, o.P, o.rhs(), args.kmax, keep_at=(5,), sample=lambda v: v[::st, ::sx],
                progress=os.path.join(REPO, 'gpurun_out', 'oracle_progress_' + os.path.basename(path) + '.json'))
        else:
            w, iters, hist = pcg(o.WT_S_W, o.P, o.rhs(), kmax=args.kmax, callback=cb)
        print('oracle PCG: %d iterations in %.1f s' % (iters, time.time() - t0))
        out = dict(J_time=args.J_time, J_space=args.J_space, problem=args.problem,
                   iters=iters, kmax=args.kmax, hist=np.array(hist),
                   w_norm=np.linalg.norm(w), w_sample=w[::st, ::sx].copy(),
                   sample_strides=np.array([st, sx]))
        if args.lean:
            for k, (nrm, smp) in kept.items():
                out['w%d_norm' % k], out['w%d_sample' % k] = nrm, smp
    if not args.no_ops:
        # the bench's vector (bench.seeded_slab): the reference's timing vector,
        # np.random.seed(128); rand(N, M) (heateq_mpi_timing.py:81-83)
        X = np.random.RandomState(128).rand(o.N, o.M)
        out['SX_sample'] = o.S(X)[::st, ::sx].copy()
        out['PX_sample'] = o.P(X)[::st, ::sx].copy()
        out['WX_sample'] = o.W(X)[::st, ::sx].copy()
    np.savez_compressed(path, **out)


if __name__ == '__main__':
    main()
