"""Space operators on device slabs, and the serial operator helpers of the
reference (counterpart of reference source/linop.py).

In the reference a "space operator" is anything SciPy can apply to an (M, k)
block with one right-hand side per column (``mat_space @ X_loc.T``,
mpi_kron.py:149).  Here a space operator acts on a device slab ``x[i, t]`` of
shape (M, ld) -- all time columns at once -- through ``apply(x, out)``.

  SpaceMatrix     a CSR matrix in HBM (SciPy CSR matrices are wrapped
                  automatically wherever the reference accepts them)
  CompositeLinOp  x -> A B x (reference linop.py:68-79)
  MultiGrid       see multigrid.py
KronLinOp / BlockDiagLinOp are the serial (single-rank, flat NumPy vector in /
out) operators of reference linop.py:6-15, 29-44, run on the device.  They are
DeviceLinearOperators: applied to a flat NumPy vector they behave as the
reference's (the vector travels to the device and back per apply); applied to a
device vector (device_vector(x, n_time): a one-rank KronVectorMPI) they return
one, sums and products of them do too, and a whole PCG solve then stays in HBM
(heateq.py: solve()).
"""
import ctypes
import os
import threading
import weakref

import numpy as np
import scipy.sparse as sp
import torch
from scipy.sparse.linalg import LinearOperator

from . import _lib
from .assembly import (tile_order_from_coords,  # noqa: F401 (re-exported)
                       tile_rows_from_coords)


class SpaceOp:
    """Protocol: ``shape`` and ``apply(x, out=None, cm=None, kind=None)`` on
    (M, ld) device slabs; must not alias x and out."""
    shape = None

    def apply(self, x, out=None, n_loc=None, **kw):
        raise NotImplementedError

    # reference-style use on host data: op @ array, one RHS per column
    def __matmul__(self, X):
        X = np.asarray(X, dtype=np.float64)
        one_d = X.ndim == 1
        Xc = X.reshape(self.shape[1], -1)
        x = _lib.to_dev(Xc)
        y = self.apply(x, n_loc=x.shape[1])
        Y = y.cpu().numpy()
        return Y.reshape(-1) if one_d else Y


_space_cache = {}


def as_space_op(mat):
    """SciPy sparse / dense matrices -> SpaceMatrix (uploaded once per matrix
    object); SpaceOps pass through."""
    if isinstance(mat, SpaceOp):
        return mat
    key = id(mat)
    hit = _space_cache.get(key)
    if hit is not None and hit[0]() is mat:
        return hit[1]
    op = SpaceMatrix(mat)
    try:
        _space_cache[key] = (weakref.ref(mat), op)
    except TypeError:
        pass
    return op


class SpaceMatrix(SpaceOp):
    def __init__(self, mat):
        if not sp.issparse(mat):
            mat = sp.csr_matrix(np.asarray(mat, dtype=np.float64))
        self.mat = sp.csr_matrix(mat)
        self.mat.sort_indices()
        self.shape = self.mat.shape
        self.dev = _lib.DeviceCSR(self.mat)
        self._ell = None
        self._hint = getattr(mat, 'stk_row_order', None)

    def _ell_form(self):
        """Sliced-ELL copy for the row-gather engine, built on first use."""
        if self._ell is None:
            m = self.mat
            order = self._hint
            if order is None:
                order = row_order_for([m], m.indptr, m.indices)
            self._ell = EllRowsMatrix(m.indptr, m.indices, m.data, None, order)
        return self._ell if self._ell.ok else None

    def apply(self, x, out=None, n_loc=None, alpha=1.0, beta=0.0, z=None,
              ld=None, **kw):
        ld = x.shape[1] if ld is None else ld  # given: x / out are column ranges of wider slabs
        n_loc = ld if n_loc is None else n_loc
        if out is None:
            out = torch.empty((self.shape[0], ld),
                              dtype=torch.float64,
                              device=x.device)
        # the row engine gives a row one lane per pair of time steps (at most 512 lanes);
        # longer rows -- the transposed slabs of MatKronIdentityMPI, whose "time" axis is
        # a rank's share of the space dofs -- take the flat CSR kernel
        ell = self._ell_form() if (ld % 2 == 0 and ld > 1 and (n_loc + 1) // 2 <= 512) else None
        if ell is not None:
            _lib.check(_lib.lib().stk_ell_spmm(
                _lib.stream(), ctypes.byref(ell.struct), n_loc, ld,
                self.shape[1], 1.0, None, _lib.ptr(x), alpha, beta,
                _lib.ptr(z), _lib.ptr(out)))
            return out
        d = self.dev
        _lib.check(_lib.lib().stk_csr_spmm(
            _lib.stream(), self.shape[0], n_loc, ld, _lib.ptr(d.indptr),
            _lib.ptr(d.indices), _lib.ptr(d.data), 1.0, None, None,
            _lib.ptr(x), alpha, beta, _lib.ptr(z), _lib.ptr(out)))
        return out


class InvLinOp(SpaceOp):
    """Direct inverse as a space operator (reference linop.py:18-26, used by
    precond='direct', heateq_mpi.py:154-157).  The factorisation is SuperLU on
    the host at setup, as in the reference.  Up to MAX_ROWS rows the device
    applies the explicit inverse with the same row-gather kernels as any other
    matrix (the sizes the reference's tests use it on: the dense inverse needs
    M^2 doubles).  Above that, what the reference's `self.inv.solve` does per apply
    -- row permutation, L and U solves, column permutation -- runs on the device on
    SuperLU's factors, level-scheduled, all time steps at once (stk_lu_solve,
    csrc/sptrsv.hip): the slab never leaves HBM.  `host_solve = True` keeps the
    round trip through SuperLU on the host of rounds 1-5 (the comparison partner
    of the tests)."""
    MAX_ROWS = 8192
    host_solve = False
    # The narrow levels near the root of the elimination tree (a few thousand rows, one
    # dependent row after the other) as ONE dense block whose triangular factors are
    # inverted at set-up: hundreds of dependent levels become two dense products
    # (csrc/sptrsv.hip).  False: every level by itself (the comparison partner of the tests).
    dense_top = True
    DENSE_TOP_MAX = 8192  # rows of the block (its two dense copies: 2 x 512 MiB at most)
    # Rows of a diagonal block that is inverted: substitution from block to block, an explicit
    # inverse inside one.  The whole block (the default: TOP_BLOCK >= n_top) is two launches per
    # solve direction -- K^-1 0.35 ms at M = 16 129 -- with 5-10 times the rounding error of
    # substitution (1.4e-15 against 2e-16), which moves the LAST entry of a converged r.Pr
    # history, 1e-13 of the first, by 5.8e-11; blocks of 256 rows keep the accuracy of
    # substitution (3.1e-11 there) for 79 launches and 0.93 ms
    # (profiles/r06_direct_parity_forms*.log, r06_direct_solve_time*.log).
    TOP_BLOCK = 8192
    n_top = 0

    def __init__(self, mat):
        mat = sp.csc_matrix(mat)
        n = mat.shape[0]
        self.shape = mat.shape
        self.dtype = np.float64
        self.lu = sp.linalg.splu(mat, options={"SymmetricMode": True},
                                 permc_spec="MMD_AT_PLUS_A")
        self._dense = None
        self._plan = None
        self._work = None
        if n <= self.MAX_ROWS:
            self._dense = SpaceMatrix(sp.csr_matrix(self.lu.solve(np.eye(n))))
            self.mat = self._dense.mat
        elif not self.host_solve and _lib.compute_device().type == 'cuda':
            self._device_plan()  # set-up work belongs to the set-up

    def _device_plan(self):
        if self._plan is None:
            L, U = sp.csr_matrix(self.lu.L), sp.csr_matrix(self.lu.U)
            for T in (L, U):
                T.sort_indices()
            i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
            f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
            args = [i32(L.indptr), i32(L.indices), f64(L.data), i32(U.indptr), i32(U.indices),
                    f64(U.data), i32(self.lu.perm_r), i32(self.lu.perm_c)]
            plan = ctypes.c_void_p()
            lib = _lib.lib()
            _lib.check(lib.stk_lu_create(self.shape[0], *[a.ctypes.data for a in args],
                                         ctypes.byref(plan)))
            # the top of the elimination tree as a dense block: the plan names the rows,
            # the two triangular blocks are inverted here (dense solves on the device)
            n_top = ctypes.c_int32()
            _lib.check(lib.stk_lu_top_rows(plan, ctypes.byref(n_top), None))
            if 0 < n_top.value <= self.DENSE_TOP_MAX and self.dense_top:
                from scipy.linalg.lapack import dtrtri
                rows = np.empty(n_top.value, dtype=np.int32)
                _lib.check(lib.stk_lu_top_rows(plan, ctypes.byref(n_top), rows.ctypes.data))
                blocks = []
                nb = min(self.TOP_BLOCK, n_top.value)
                for T, lower, unit in ((L, 1, 1), (U, 0, 0)):
                    # the block as a dense matrix with its DIAGONAL BLOCKS of nb rows inverted
                    # (LAPACK's triangular inverse on the host, at set-up like the
                    # factorisation itself; hipBLAS' trsm fails to allocate on this image)
                    dense = np.ascontiguousarray(T[rows][:, rows].toarray())
                    for a in range(0, n_top.value, nb):
                        e = min(a + nb, n_top.value)
                        inv, info = dtrtri(np.asfortranarray(dense[a:e, a:e]), lower=lower, unitdiag=unit)
                        assert info == 0, info
                        dense[a:e, a:e] = inv
                    blocks.append(_lib.to_dev(dense))
                _lib.check(lib.stk_lu_set_top_inverse(plan, _lib.ptr(blocks[0]), _lib.ptr(blocks[1]), nb))
                self.n_top = int(n_top.value)
            self._plan = plan
        return self._plan

    def levels(self):
        """(dependency levels of the L solve, of the U solve, kernel launches per apply)."""
        out = [ctypes.c_int32() for _ in range(3)]
        _lib.check(_lib.lib().stk_lu_info(self._device_plan(), *[ctypes.byref(v) for v in out]))
        return tuple(v.value for v in out)

    def apply(self, x, out=None, n_loc=None, **kw):
        if self._dense is not None:
            return self._dense.apply(x, out=out, n_loc=n_loc, **kw)
        M, ld = x.shape
        n_loc = ld if n_loc is None else n_loc
        if out is None:
            out = torch.empty_like(x)
        lib = _lib.lib()
        if not self.host_solve:
            if self._work is None or self._work.shape != x.shape or self._work.device != x.device:
                self._work = torch.empty_like(x)
            _lib.check(lib.stk_lu_solve(self._device_plan(), _lib.stream(), n_loc, ld, _lib.ptr(x),
                                        _lib.ptr(out), _lib.ptr(self._work)))
            return out
        host = np.empty((n_loc, M))
        _lib.check(lib.stk_slab_download(_lib.stream(), M, n_loc, ld, _lib.ptr(x),
                                         host.ctypes.data))
        sol = np.ascontiguousarray(self.lu.solve(np.ascontiguousarray(host.T)).T)
        _lib.check(lib.stk_slab_upload(_lib.stream(), M, n_loc, ld,
                                       sol.ctypes.data, _lib.ptr(out)))
        return out

    def __del__(self):
        try:
            if self._plan is not None:
                _lib.lib().stk_lu_destroy(self._plan)
        except Exception:
            pass


class CompositeLinOp(SpaceOp):
    """x -> A B x, applied right to left (reference linop.py:68-79)."""
    def __init__(self, linops):
        self.linops = [as_space_op(op) for op in linops]
        self.shape = (self.linops[0].shape[0], self.linops[-1].shape[1])
        self.dtype = np.float64

    def apply(self, x, out=None, n_loc=None, **kw):
        y = x
        for k, op in enumerate(reversed(self.linops)):
            last = k == len(self.linops) - 1
            y = op.apply(y, out=out if last else None, n_loc=n_loc, **kw)
        return y


def union_pattern(mats):
    """One CSR pattern containing the patterns of all `mats`, and every
    matrix's values expanded onto it (zeros where it has no entry).  The fused
    kernels walk the pattern once for all matrices.

    The patterns are added with weights 2^k, so that an entry of the sum says
    which matrices own it; a matrix's values then fall onto the union in their own
    (row-major, column-sorted) order -- no search."""
    mats = [sp.csr_matrix(m) for m in mats]
    for m in mats:
        if not m.has_sorted_indices:
            m.sort_indices()
        m.sum_duplicates()
    assert len(mats) <= 30
    # the same large matrices are united more than once during a set-up (the Kronecker
    # plan of S takes (M_x, A_x), the preconditioner family's finest level (A_x, M_x)):
    # the last result is kept, keyed by the matrices' arrays, and handed out in the
    # caller's order
    keys = [(m.indptr.ctypes.data, m.indices.ctypes.data, m.data.ctypes.data, m.nnz) for m in mats]
    if len(mats) == 1 or sum(m.nnz for m in mats) <= 1000000:
        return _union_pattern(mats)
    ident = frozenset(keys)
    with _union_lock:
        slot = _union_slots.get(ident)
        if slot is None:
            _union_slots.clear()  # one entry: these arrays are large
            slot = _union_slots[ident] = {'lock': threading.Lock()}
    with slot['lock']:  # whoever comes second (the plans are built side by side) waits
        if 'value' not in slot:
            slot['value'] = (keys, _union_pattern(mats), mats)  # mats: the keys hold addresses
    h_keys, (h_ptr, h_idx, h_vals), _ = slot['value']
    return h_ptr, h_idx, [h_vals[h_keys.index(k)] for k in keys]


_union_slots, _union_lock = {}, threading.Lock()


def forget_union_pattern():
    """Drops the kept union pattern (HeatEquationMPI calls it when its plans are built:
    the arrays are as large as the matrices)."""
    with _union_lock:
        _union_slots.clear()


# matrices with at least this many entries together are united by libstk's host threads
UNION_ON_HOST_THREADS = 200000


def _union_pattern_libstk(mats):
    """The same arrays from stk_csr_union_count / _fill (csrc/mesh_refine.hip): the rows
    merged on the host threads of the library, no interpreter lock held (the SciPy form
    below adds weighted patterns and scatters the values; 0.06 s alone at config 3 and
    0.17 s beside the other planners, on the critical path of two of them)."""
    lib = _lib.lib()
    k, n = len(mats), mats[0].shape[0]
    assert all(m.shape == mats[0].shape for m in mats)
    ptrs = [np.ascontiguousarray(m.indptr) for m in mats]
    idxs = [np.ascontiguousarray(m.indices) for m in mats]
    data = [np.ascontiguousarray(m.data, dtype=np.float64) for m in mats]
    table = lambda arrays: (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrays])
    indptr = np.empty(n + 1, dtype=np.int32)
    _lib.check(lib.stk_csr_union_count(n, k, table(ptrs), table(idxs), indptr.ctypes.data))
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    vals = [np.empty(nnz) for _ in mats]
    _lib.check(lib.stk_csr_union_fill(n, k, table(ptrs), table(idxs), table(data), indptr.ctypes.data,
                                      indices.ctypes.data, table(vals)))
    return indptr, indices, vals


def _union_pattern(mats):
    if len(mats) == 1:  # nothing to unite
        m = mats[0]
        return (np.asarray(m.indptr, dtype=np.int32), np.asarray(m.indices, dtype=np.int32),
                [np.asarray(m.data, dtype=np.float64)])
    if (sum(m.nnz for m in mats) >= UNION_ON_HOST_THREADS
            and all(m.indices.dtype == np.int32 and m.indptr.dtype == np.int32 for m in mats)):
        return _union_pattern_libstk(mats)
    pat = None
    for k, m in enumerate(mats):
        own = sp.csr_matrix((np.full(m.nnz, float(1 << k)), m.indices, m.indptr),
                            shape=m.shape)
        pat = own if pat is None else pat + own
    pat = sp.csr_matrix(pat)
    pat.sort_indices()
    owners = pat.data.astype(np.int64)
    vals = []
    for k, m in enumerate(mats):
        full = np.zeros(pat.nnz)
        full[((owners >> k) & 1) == 1] = m.data
        vals.append(full)
    return np.asarray(pat.indptr, dtype=np.int32), np.asarray(pat.indices, dtype=np.int32), vals


def row_order_for(mats, indptr, indices):
    """Processing order of the rows for the gather kernels: the hint attached
    by the assembly (`stk_row_order`, a mesh-tile order) if there is one,
    reverse Cuthill-McKee of the pattern for large anonymous matrices, else
    None (index order)."""
    for m in mats:
        order = getattr(m, 'stk_row_order', None)
        if order is not None:
            return np.asarray(order, dtype=np.int32)
    n = len(indptr) - 1
    if n < 8192:
        return None
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    pat = sp.csr_matrix((np.ones(len(indices)), indices, indptr),
                        shape=(n, int(indices.max()) + 1 if len(indices) else n))
    if pat.shape[0] != pat.shape[1]:
        return None
    return reverse_cuthill_mckee(pat, symmetric_mode=True).astype(np.int32)


def permute_rows(indptr, indices, vals, order):
    """CSR rows listed in `order`: returns (indptr, indices, vals, row_ids)
    such that CSR row k of the result is row order[k] of the input."""
    if order is None:
        return indptr, indices, vals, None
    counts = np.diff(indptr)[order]
    new_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    starts = indptr[:-1][order]
    gather = np.repeat(starts - new_ptr[:-1], counts) + np.arange(new_ptr[-1])
    return (new_ptr, indices[gather].astype(np.int32),
            [v[gather] for v in vals], np.asarray(order, dtype=np.int32))


class EllRowsMatrix:
    """One matrix (values va, optionally a second value array vm on the same
    pattern) in the sliced-ELL form of the row-gather engine (stk_ell_rows,
    include/stk.h).  `order` lists the rows in processing order; `diag` makes
    the Gauss-Seidel copy (diagonal entries in their own arrays, off-diagonal
    entries in the slots: stk_ell_rows.diag_free); `dia_values=True` keeps the whole
    row in the slots and records the diagonal beside it; `earlier_group` (a group
    index per row) keeps only the entries whose column lies in an earlier group
    than the row (the zero-start copies of csrc/mg.hip).  `ok` is False when a row
    has more entries than the largest instantiated slot count.

    On a GPU the copy is made by libstk from the CSR arrays, which are uploaded
    once per matrix (stk_ell_from_csr); without one (CPU tests of the planner) by
    the NumPy code below, which produces the same arrays."""
    SLOTS = (2, 4, 5, 6, 7, 9, 12, 16, 20)
    _tls = threading.local()  # per-thread upload cache: (key, device arrays, referents)

    def __init__(self, indptr, indices, va, vm=None, order=None, n_cols=None,
                 diag=False, pad_col=0, dia_values=None, earlier_group=None,
                 kept_counts=None):
        n = len(indptr) - 1
        order = np.arange(n, dtype=np.int64) if order is None else np.asarray(
            order, dtype=np.int64)
        counts = np.diff(indptr).astype(np.int64)
        if earlier_group is not None:
            counts = kept_counts  # entries per row that pass the filter (shared by the groups' copies)
            if counts is None:
                rows_of = np.repeat(np.arange(n), np.diff(indptr))
                keep = earlier_group[indices] < earlier_group[rows_of]
                counts = np.bincount(rows_of[keep], minlength=n)
        elif diag:
            counts = counts - 1  # every row has its diagonal entry (checked below)
        # slot count of THIS copy: widest of the rows it lists
        kmax = int(counts[order].max()) if len(order) else 1
        self.ok = kmax <= self.SLOTS[-1]
        if not self.ok:
            return
        K = next(k for k in self.SLOTS if k >= max(kmax, 1))
        npos = len(order)
        self.K, self.n_pos, self.n_rows = K, npos, n
        want_dia = diag or dia_values is not None
        if _lib.compute_device().type == 'cuda':
            self._build_on_device(indptr, indices, va, vm, order, K, diag, pad_col,
                                  want_dia, earlier_group)
        else:
            self._build_with_numpy(indptr, indices, va, vm, order, K, diag, pad_col,
                                   want_dia, earlier_group)
        self.row_ids = _lib.to_dev(order.astype(np.int32))
        self.struct = _lib.EllRows(npos, n, K, _lib.ptr(self.idx),
                                   _lib.ptr(self.va), _lib.ptr(self.vm),
                                   _lib.ptr(self.row_ids),
                                   _lib.ptr(self.dia_a), _lib.ptr(self.dia_m),
                                   int(diag))

    @classmethod
    def _uploaded(cls, indptr, indices, va, vm):
        """The CSR arrays on the device, uploaded once per matrix and thread (the
        copies of one level -- a, fwd, bwd, the zero-start ones -- share them)."""
        key = (id(indptr), id(indices), id(va), id(vm))
        held = getattr(cls._tls, 'dev', None)
        if held is None or held[0] != key:
            dev = (_lib.to_dev(np.asarray(indptr, dtype=np.int32)),
                   _lib.to_dev(np.asarray(indices, dtype=np.int32)),
                   _lib.to_dev(np.asarray(va, dtype=np.float64)),
                   None if vm is None else _lib.to_dev(np.asarray(vm, dtype=np.float64)))
            held = (key, dev, (indptr, indices, va, vm))  # the key holds ids: keep the referents
            cls._tls.dev = held
        return held[1]

    def _build_on_device(self, indptr, indices, va, vm, order, K, diag, pad_col,
                         want_dia, earlier_group):
        d_ptr, d_idx, d_va, d_vm = self._uploaded(indptr, indices, va, vm)
        dev, npos = d_ptr.device, len(order)
        f64 = dict(dtype=torch.float64, device=dev)
        self.idx = torch.empty((npos, K), dtype=torch.int32, device=dev)
        self.va = torch.empty((npos, K), **f64)
        self.vm = None if vm is None else torch.empty((npos, K), **f64)
        self.dia_a = torch.empty(npos, **f64) if want_dia else None
        self.dia_m = torch.empty(npos, **f64) if want_dia and vm is not None else None
        d_order = _lib.to_dev(order.astype(np.int32))
        d_grp = None
        if earlier_group is not None:  # uploaded once for the copies of all groups
            held = getattr(self._tls, 'grp', None)
            if held is None or held[0] is not earlier_group:
                held = (earlier_group, _lib.to_dev(np.asarray(earlier_group, dtype=np.int32)))
                self._tls.grp = held
            d_grp = held[1]
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().stk_ell_from_csr(
            _lib.stream(), npos, K, _lib.ptr(d_order), _lib.ptr(d_ptr), _lib.ptr(d_idx),
            _lib.ptr(d_va), _lib.ptr(d_vm), int(diag), _lib.ptr(d_grp), int(pad_col),
            _lib.ptr(self.idx), _lib.ptr(self.va), _lib.ptr(self.vm),
            _lib.ptr(self.dia_a), _lib.ptr(self.dia_m), _lib.ptr(overflow)))
        self._overflow = overflow  # read lazily: no synchronisation per copy
        self._keep = (d_order, d_grp)

    def check(self):
        """Synchronises and verifies that no listed row was longer than K (the
        host sizes K from the row lengths, so this can only fail for a matrix
        without a diagonal entry in a Gauss-Seidel copy)."""
        ov = getattr(self, '_overflow', None)
        if ov is not None and int(ov.item()) != 0:
            raise _lib.StkError('EllRowsMatrix: a row has %d entries for %d slots'
                                % (int(ov.item()), self.K))

    @staticmethod
    def check_all(copies):
        """check() of several copies with ONE synchronisation (the plan builders
        call it once per level, after all copies of the level are queued)."""
        flags = [c._overflow for c in copies if getattr(c, '_overflow', None) is not None]
        if flags and int(torch.stack(flags).max().item()) != 0:
            for c in copies:
                c.check()

    def _build_with_numpy(self, indptr, indices, va, vm, order, K, diag, pad_col,
                          want_dia, earlier_group):
        n = len(indptr) - 1
        rows_of = np.repeat(np.arange(n), np.diff(indptr))
        on = indices == rows_of
        dia = None
        if want_dia:
            assert np.array_equal(np.bincount(rows_of[on], minlength=n),
                                  np.ones(n, dtype=np.int64)), 'matrix lacks a diagonal entry'
            dia = (np.asarray(va)[on], None if vm is None else np.asarray(vm)[on])
        keep = np.ones(len(indices), dtype=bool)
        if diag:
            keep &= ~on
        if earlier_group is not None:
            keep &= earlier_group[indices] < earlier_group[rows_of]
        counts = np.bincount(rows_of[keep], minlength=n)
        ptr = np.concatenate([[0], np.cumsum(counts)])
        k_idx, k_rows = np.asarray(indices)[keep], rows_of[keep]
        slot = np.arange(len(k_idx)) - np.repeat(ptr[:-1], counts)
        fits = slot < K
        idx0 = np.full((n, K), max(pad_col, 0), dtype=np.int32)
        if pad_col < 0:  # the row's first kept column, or the row itself
            first = np.arange(n, dtype=np.int32)
            has = counts > 0
            first[has] = k_idx[ptr[:-1][has]]
            idx0[:] = first[:, None]
        idx0[k_rows[fits], slot[fits]] = k_idx[fits]
        ea0 = np.zeros((n, K))
        ea0[k_rows[fits], slot[fits]] = np.asarray(va)[keep][fits]
        self.idx = _lib.to_dev(idx0[order])
        self.va = _lib.to_dev(ea0[order])
        self.vm = None
        if vm is not None:
            em0 = np.zeros((n, K))
            em0[k_rows[fits], slot[fits]] = np.asarray(vm)[keep][fits]
            self.vm = _lib.to_dev(em0[order])
        self.dia_a = self.dia_m = None
        if dia is not None:
            self.dia_a = _lib.to_dev(dia[0][order])
            if dia[1] is not None:
                self.dia_m = _lib.to_dev(dia[1][order])


class EllMatrices:
    """Several matrices on one shared pattern in the sliced-ELL form of
    ``stk_kron_ell_apply`` (include/stk.h), resident on the device: K slots per
    row, rows in processing order, overflow CSR for rows longer than K."""
    MAXK = 16
    SLOTS = (5, 7, 9, 12, 16)  # instantiated in csrc/kron_ell.hip
    _shared, _shared_lock = [], threading.Lock()

    @classmethod
    def shared(cls, mats, order_hints=()):
        """The plan of exactly these matrix objects, built once: S and the
        metric's operator both stream (M_x, A_x).  Keyed by the CSR arrays (not
        their values: matrices are not modified in place anywhere); the few most
        recent plans are kept together with their matrices, so the arrays stay
        alive."""
        mats = [sp.csr_matrix(m) for m in mats]  # shares the arrays of a CSR input
        key = tuple((m.indptr.ctypes.data, m.indices.ctypes.data,
                     m.data.ctypes.data, m.nnz) for m in mats)
        with cls._shared_lock:
            for k, _, plan in cls._shared:
                if k == key:
                    return plan
        plan = cls(mats, order_hints)
        with cls._shared_lock:
            cls._shared.append((key, list(mats), plan))
            del cls._shared[:-4]
        return plan

    def __init__(self, mats, order_hints=()):
        indptr, indices, vals = union_pattern(mats)
        order = row_order_for(list(order_hints) + list(mats), indptr, indices)
        indptr, indices, vals, row_ids = permute_rows(indptr, indices, vals,
                                                      order)
        M = len(indptr) - 1
        counts = np.diff(indptr)
        kmax = int(counts.max()) if M else 1
        K = next((k for k in self.SLOTS if k >= kmax), self.MAXK)
        own = row_ids if row_ids is not None else np.arange(M, dtype=np.int32)
        pos = np.repeat(np.arange(M), counts)
        slot = np.arange(len(indices)) - np.repeat(indptr[:-1], counts)
        main = slot < K
        # (flat one-dimensional scatters: several times faster than [rows, slots]
        # index pairs on seven million entries)
        flat = pos * K + slot
        whole = bool(main.all())
        if not whole:
            flat = flat[main]
        ell_idx = np.repeat(own.astype(np.int32)[:, None], K, axis=1)
        ell_idx.reshape(-1)[flat] = indices if whole else indices[main]
        ell_vals = []
        for v in vals:
            e = np.zeros((M, K))
            e.reshape(-1)[flat] = v if whole else v[main]
            ell_vals.append(e)
        self.M, self.K = M, K
        self.nnz = len(indices)
        self.nnz_terms = [int(sp.csr_matrix(m).nnz) for m in mats]
        self.ell_idx = _lib.to_dev(ell_idx)
        self.ell_vals = [_lib.to_dev(e) for e in ell_vals]
        self.row_ids = None if row_ids is None else _lib.to_dev(row_ids)
        self.ovf_indptr = self.ovf_indices = None
        self.ovf_vals = [None] * len(vals)
        if not main.all():
            rest = ~main
            oc = np.bincount(pos[rest], minlength=M)
            self.ovf_indptr = _lib.to_dev(
                np.concatenate([[0], np.cumsum(oc)]).astype(np.int32))
            self.ovf_indices = _lib.to_dev(indices[rest].astype(np.int32))
            self.ovf_vals = [_lib.to_dev(v[rest]) for v in vals]
        self.pattern = _lib.EllPattern(M, K, _lib.ptr(self.ell_idx),
                                       _lib.ptr(self.row_ids),
                                       _lib.ptr(self.ovf_indptr),
                                       _lib.ptr(self.ovf_indices))
        self._pack_args = (M, K, ell_idx, ell_vals, self.row_ids,
                           not main.all(), counts, own)
        self._packed, self._packed_lock = {}, threading.Lock()

    # Slabs of fewer time steps than this keep one row per slot row.  Round 2 measured
    # pairs to win from 32 steps on and to lose below (profiles/r02_slab_shapes2.log) and
    # set 24; on today's kernel they win on every slab shape of a 2-, 4- and 8-rank run
    # (profiles/r05_slab_shapes_J9.log, _J10.log: 9 steps 0.053 -> 0.050 ms, 17 steps
    # 0.086 -> 0.082 ms at J_space = 9, 0.429 -> 0.359 ms at J_space = 10).  Pairs with
    # EXPLICIT values (no dictionary) were only measured from 33 steps on
    # (profiles/r03_ab_jitter_J8_33.log): below 24 such matrices keep the plain form.
    PAIR_MIN_STEPS = 8
    EXPLICIT_PAIR_MIN_STEPS = 24

    @property
    def packed(self):
        """The packed form long slabs use (row pairs where the matrices allow)."""
        return self.packed_variant(PackedEllMatrices.ROWS_PER_UNIT)

    def packed_for(self, n_loc):
        """The packed form for a slab of n_loc time steps."""
        rows = PackedEllMatrices.ROWS_PER_UNIT if n_loc >= self.PAIR_MIN_STEPS else 1
        form = self.packed_variant(rows)
        if form.ok and form.explicit and n_loc < self.EXPLICIT_PAIR_MIN_STEPS:
            form = self.packed_variant(1)
        return form

    def packed_variant(self, rows_per_unit):
        """The packed form with 1 or 2 matrix rows per slot row (falls back to 1
        where rows cannot share slot rows); built on first use."""
        with self._packed_lock:
            if rows_per_unit not in self._packed:
                packed = PackedEllMatrices(*self._pack_args, rows_per_unit=rows_per_unit)
                if not packed.ok and rows_per_unit > 1:
                    packed = self._packed.get(1) or PackedEllMatrices(
                        *self._pack_args, rows_per_unit=1)
                    self._packed.setdefault(1, packed)
                self._packed[rows_per_unit] = packed
            return self._packed[rows_per_unit]

    def _terms(self, specs, ghosts):
        terms = (_lib.KronEllTerm * len(specs))()
        for t, (tri, k, x, lo, hi) in zip(terms, specs):
            t.tri, t.ell_vals = _lib.ptr(tri), _lib.ptr(self.ell_vals[k])
            t.ovf_vals = _lib.ptr(self.ovf_vals[k])
            t.x = _lib.ptr(x)
            t.x_lo = _lib.ptr(lo) if ghosts else None
            t.x_hi = _lib.ptr(hi) if ghosts else None
        return terms

    def apply(self, specs, n_loc, ld, beta, out):
        """y = beta*y + sum over specs (tri, matrix index, x, x_lo, x_hi)."""
        _lib.check(_lib.lib().stk_kron_ell_apply(
            _lib.stream(), ctypes.byref(self.pattern), n_loc, ld, len(specs),
            self._terms(specs, True), beta, _lib.ptr(out)))

    def apply_local(self, specs, n_loc, ld, beta, out):
        """The part of `apply` that needs no ghost rows (x_lo, x_hi ignored):
        can run while the halo exchange is in flight.  With beta = 0 when
        apply_ghost is to follow."""
        _lib.check(_lib.lib().stk_kron_ell_apply(
            _lib.stream(), ctypes.byref(self.pattern), n_loc, ld, len(specs),
            self._terms(specs, False), beta, _lib.ptr(out)))

    def apply_ghost(self, specs, n_loc, ld, out):
        """apply = apply_local (beta = 0), then apply_ghost once the exchange has
        completed: the first and last local time step are recomputed with the
        received rows, in the order of operations of the main kernel, and
        overwrite what apply_local left there -- the result does not depend on
        where the slabs are cut (stk_kron_ell_ghost_apply)."""
        if any(lo is not None or hi is not None for _, _, _, lo, hi in specs):
            _lib.check(_lib.lib().stk_kron_ell_ghost_apply(
                _lib.stream(), ctypes.byref(self.pattern), n_loc, ld,
                len(specs), self._terms(specs, True), _lib.ptr(out)))


def time_factor_steps(tri_host):
    """[t_begin, t_end): the local time steps at which a time factor given by its
    three diagonals (`tri_host`: (3, n_loc) sub / main / super, as the kernels take
    them) reads its input -- column t is reached through sub[t + 1], dia[t] and
    super[t - 1].  None for an identity factor (tri_host None): every step."""
    if tri_host is None:
        return None
    t = np.asarray(tri_host)
    n = t.shape[1]
    used = t[1] != 0.0
    used[:-1] |= t[0, 1:] != 0.0
    used[1:] |= t[2, :-1] != 0.0
    idx = np.flatnonzero(used)
    return (0, 0) if len(idx) == 0 else (int(idx[0]), int(idx[-1]) + 1)


def _few_unique(v, limit):
    """np.unique(v, return_inverse=True) for an int64 array that is expected to
    hold at most `limit` distinct values: candidates from a sample, then binary
    search of everything (no sort of the whole array).  None if there are more."""
    v = np.ascontiguousarray(v).reshape(-1)
    u = np.unique(v[::max(1, len(v) // 65536)])
    while len(u) <= limit:
        pos = np.minimum(np.searchsorted(u, v), len(u) - 1)
        miss = u[pos] != v
        if not miss.any():
            return u, pos
        u = np.union1d(u, np.unique(v[miss][:1 << 20]))
    return None


def _small_unique(v, bound):
    """np.unique(v, return_inverse=True) for non-negative integers below a small
    `bound` (a table instead of a sort)."""
    present = np.zeros(bound, dtype=bool)
    present[v] = True
    u = np.flatnonzero(present)
    lut = np.cumsum(present) - 1
    return u, lut[v]


class PackedEllMatrices:
    """The matrices of an EllMatrices plan in the packed form of
    ``stk_kron_pack_apply`` (include/stk.h): one 32-bit word per slot,
    ``code << col_bits | column``, plus the dictionary of the distinct value
    tuples of the union pattern.  With ``rows_per_unit`` = 2..4, rows that follow
    each other in the processing order and share columns (mesh neighbours) are
    served by ONE slot row listing the union of their columns.  ``ok`` is False
    when the plan does not fit (overflow rows, more distinct tuples than the
    free bits can name, a slot count without a pair instantiation); the caller
    then keeps the plain form."""
    MAX_CODES = 512  # dictionary entries (codes x rows per unit) in LDS next to the exchange buffers
    ROWS_PER_UNIT = int(os.environ.get('STK_PACK_ROWS', '2'))

    def __init__(self, M, K, ell_idx, ell_vals, row_ids, has_overflow,
                 counts=None, own=None, rows_per_unit=1):
        self.ok = False
        self.explicit = False
        self.rows_per_unit = rows_per_unit
        if has_overflow or M < 1:
            return
        K_out = K
        if rows_per_unit > 1:
            K_out = int(_lib.lib().stk_pack_unit_slots(K, rows_per_unit))
            if K_out == 0 or counts is None:
                return
        col_bits = max(1, int(M - 1).bit_length())
        self._no_dictionary = False
        built = self._with_dictionary(M, K, K_out, col_bits, ell_idx, ell_vals,
                                      counts, own, rows_per_unit)
        if (built is None and self._no_dictionary and rows_per_unit == 2
                and self.EXPLICIT_PAIRS):
            # values that do not repeat (an unstructured mesh): no dictionary, but
            # neighbouring rows still share COLUMNS -- pairs with explicit values
            built = self._with_explicit_values(M, K, K_out, ell_idx, ell_vals,
                                               counts, own)
        if built is None:
            return
        cols, codes, table, unit_rows, vals = built
        self.ok = True
        self.explicit = vals is not None
        self.rows_per_unit = rows_per_unit
        self.n_units, self.K = cols.shape
        self.M, self.col_bits = M, col_bits
        self.n_mats = len(ell_vals)
        self.row_ids = row_ids if unit_rows is None else _lib.to_dev(unit_rows)
        self.dict = self.vals = None
        if vals is not None:
            self.n_codes = 1
            self.slots = _lib.to_dev(cols.astype(np.uint32).view(np.int32).reshape(cols.shape))
            # the kernel reads the values of a call's terms side by side: one device
            # array per combination of matrices (built on first use, _pattern_for)
            self._host_vals, self._vals_for = vals, {}
        else:
            self.n_codes = len(table)
            slots = (codes.reshape(-1).astype(np.uint32) << np.uint32(col_bits)
                     ) | cols.reshape(-1).astype(np.uint32)
            self.slots = _lib.to_dev(slots.view(np.int32).reshape(cols.shape))
            # dict[m][code][row of the unit]
            self.dict = _lib.to_dev(np.ascontiguousarray(
                table.reshape(len(table), rows_per_unit, self.n_mats).transpose(
                    2, 0, 1)).view(np.float64))
        self.pattern = _lib.PackPattern(M, self.K, col_bits, self.n_codes,
                                        self.n_mats, rows_per_unit,
                                        self.n_units, _lib.ptr(self.slots),
                                        _lib.ptr(self.row_ids),
                                        _lib.ptr(self.dict), _lib.ptr(self.vals))

    # Row pairs with explicit values for matrices without a dictionary (False:
    # such matrices keep the one-row plain form, stk_kron_ell_apply).
    EXPLICIT_PAIRS = True
    MATCH_WINDOW = 8192  # positions of the processing order a row's partner may come from

    def _with_dictionary(self, M, K, K_out, col_bits, ell_idx, ell_vals, counts,
                         own, rows_per_unit):
        """(columns, codes, dictionary, unit rows, None) of the packed form, or
        None when the values do not fit a dictionary."""
        # distinct value tuples, by bit pattern (+0.0 and -0.0 stay distinct):
        # one 1-D unique per matrix, then one over the combined codes
        codes, table = np.zeros(M * K, dtype=np.int64), None
        for e in ell_vals:
            found = _few_unique(e.reshape(-1).view(np.int64), self.MAX_CODES)
            if found is None:
                self._no_dictionary = True
                return None
            u, inv = found
            uc, codes = _small_unique(codes * len(u) + inv, self.MAX_CODES * len(u))
            if len(uc) > self.MAX_CODES:
                self._no_dictionary = True
                return None
            col = u[uc % len(u)][:, None]
            table = col if table is None else np.hstack(
                [table[uc // len(u)], col])
        uniq = table  # (n_codes, n_mats) bit patterns
        cols = ell_idx.reshape(M, K)
        unit_rows = None
        if rows_per_unit > 1:
            grouped = self._group_rows(M, K, K_out, rows_per_unit, cols,
                                       codes.reshape(M, K), uniq,
                                       np.asarray(counts), np.asarray(own))
            if grouped is None:
                return None
            cols, codes, uniq, unit_rows = grouped
        if len(uniq) > (1 << (32 - col_bits)):
            return None
        return cols, codes, uniq, unit_rows, None

    def _with_explicit_values(self, M, K, K_out, ell_idx, ell_vals, counts, own):
        """(columns, None, None, unit rows, values) of the pair form without a
        dictionary: every entry is its own "code" (its position in the ELL arrays),
        the pairing is the one of the dictionary form (stk_pack_group_rows), and the
        values of both rows of every slot are stored next to the columns:
        vals[unit][slot][row][matrix], zero where a row has no entry in a column."""
        c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        rp = 2
        entry = np.arange(1, M * K + 1, dtype=np.int32).reshape(M, K)  # 0 = "no entry"
        counts, cols, own = c32(counts), c32(ell_idx.reshape(M, K)), c32(own)
        # the processing order of an unstructured mesh does not put neighbours next
        # to each other: move one matching neighbour behind every row first
        # (locality window: a few mesh tiles)
        perm = np.empty(M, dtype=np.int32)
        _lib.check(_lib.lib().stk_pack_match_order(
            M, K, counts.ctypes.data, cols.ctypes.data, own.ctypes.data, K_out,
            self.MATCH_WINDOW, perm.ctypes.data))
        counts, cols, own, entry = c32(counts[perm]), c32(cols[perm]), c32(own[perm]), c32(entry[perm])
        ucols = np.empty((M, K_out), dtype=np.int32)
        ucodes = np.empty((M, K_out, rp), dtype=np.int32)
        urows = np.empty((M, rp), dtype=np.int32)
        n_units = ctypes.c_int32()
        _lib.check(_lib.lib().stk_pack_group_rows(
            M, K, counts.ctypes.data, cols.ctypes.data, entry.ctypes.data,
            own.ctypes.data, 0, rp, K_out, ctypes.byref(n_units),
            ucols.ctypes.data, ucodes.ctypes.data, urows.ctypes.data))
        U = n_units.value
        if U > 0.95 * M:  # hardly any rows share a unit
            return None
        ucodes = ucodes[:U]
        vals = np.empty((U, K_out, rp, len(ell_vals)))
        for m, e in enumerate(ell_vals):
            flat = np.concatenate([[0.0], e.reshape(-1)])
            vals[..., m] = flat[ucodes]
        return (ucols[:U].astype(np.int64), None, None,
                np.ascontiguousarray(urows[:U]), np.ascontiguousarray(vals))

    def _group_rows(self, M, K, K_out, rp, cols, codes, uniq, counts, own):
        """Units of up to `rp` rows that follow each other in the processing
        order and whose union of columns fits K_out slots (greedy, left to right:
        stk_pack_group_rows, host code of libstk shared with the C planner).
        Returns (columns, codes, dictionary, rows) of the units; a dictionary row
        holds the value tuples of the unit's rows one after the other."""
        n_mats = uniq.shape[1]
        zero = np.flatnonzero((uniq == 0).all(axis=1))  # code of "no entry" (+0.0 everywhere)
        if len(zero) == 0:
            uniq = np.vstack([uniq, np.zeros((1, n_mats), dtype=uniq.dtype)])
            zero = [len(uniq) - 1]
        zero, n1 = int(zero[0]), len(uniq)
        c32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        counts, cols, codes, own = c32(counts), c32(cols), c32(codes), c32(own)
        ucols = np.empty((M, K_out), dtype=np.int32)
        ucodes = np.empty((M, K_out, rp), dtype=np.int32)
        urows = np.empty((M, rp), dtype=np.int32)
        n_units = ctypes.c_int32()
        _lib.check(_lib.lib().stk_pack_group_rows(
            M, K, counts.ctypes.data, cols.ctypes.data, codes.ctypes.data,
            own.ctypes.data, zero, rp, K_out, ctypes.byref(n_units),
            ucols.ctypes.data, ucodes.ctypes.data, urows.ctypes.data))
        U = n_units.value
        if U > 0.95 * M:  # hardly any rows share a unit: not worth the wider slot rows
            return None
        ucols, ucodes, urows = ucols[:U], ucodes[:U], urows[:U]
        combined = np.zeros((U, K_out), dtype=np.int64)
        for j in range(rp):
            combined = combined * n1 + ucodes[:, :, j]
        if n1**rp <= 1 << 24:
            ucomb, ucode = _small_unique(combined.reshape(-1), n1**rp)
        else:
            ucomb, ucode = np.unique(combined.reshape(-1), return_inverse=True)
        if len(ucomb) * rp > 2 * self.MAX_CODES:
            return None
        parts, rest = [], ucomb
        for j in range(rp):
            parts.append(uniq[rest % n1])
            rest = rest // n1
        unit_dict = np.hstack(parts[::-1])
        return (ucols.astype(np.int64), ucode.reshape(U, K_out), unit_dict,
                np.ascontiguousarray(urows))

    def _terms(self, specs):
        terms = (_lib.KronPackTerm * len(specs))()
        explicit = self.explicit
        for j, (t, (tri, k)) in enumerate(zip(terms, specs)):
            t.tri, t.mat = _lib.ptr(tri), (j if explicit else k)
        return terms

    def _pattern_for(self, specs):
        """The pattern a call streams: the plan's own, or -- explicit values -- the
        one whose value array lists exactly the matrices of the call's terms, in
        term order (stk_pack_pattern.vals)."""
        if not self.explicit:
            return self.pattern
        mats = tuple(k for _, k in specs)
        hit = self._vals_for.get(mats)
        if hit is None:
            vals = _lib.to_dev(np.ascontiguousarray(self._host_vals[..., list(mats)]))
            pat = _lib.PackPattern(self.M, self.K, self.col_bits, 1, len(mats),
                                   self.rows_per_unit, self.n_units,
                                   _lib.ptr(self.slots), _lib.ptr(self.row_ids),
                                   None, _lib.ptr(vals))
            hit = self._vals_for[mats] = (vals, pat)
        return hit[1]

    def apply_ghost(self, specs, x, x_lo, x_hi, n_loc, ld, out):
        """Completes the first and last local time step after `apply` ran without
        the ghost steps (ghosts=None, beta = 0) while the halo was in flight: both
        are recomputed from `x` and the received rows x_lo / x_hi (contiguous, or
        None) in the arithmetic order of the one-pass form and overwrite what the
        pass left, so that the result does not depend on where the slabs are cut
        (stk_kron_pack_ghost_apply)."""
        if x_lo is None and x_hi is None:
            return
        _lib.check(_lib.lib().stk_kron_pack_ghost_apply(
            _lib.stream(), ctypes.byref(self._pattern_for(specs)), n_loc, ld, len(specs),
            self._terms(specs), _lib.ptr(x), _lib.ptr(x_lo), _lib.ptr(x_hi), _lib.ptr(out)))

    def apply_boundary(self, specs, records, ghosts, has_lo, has_hi, n_loc, ld, out):
        """apply_ghost from compact operands (stk_kron_pack_boundary_apply): `records` (M, 4) as
        KronVectorMPI.communicate_bdr(records=True) leaves them, `ghosts` (M, 2) interleaved
        received rows; the same doubles, 1.5-2 x faster."""
        if not (has_lo or has_hi):
            return
        _lib.check(_lib.lib().stk_kron_pack_boundary_apply(
            _lib.stream(), ctypes.byref(self._pattern_for(specs)), n_loc, ld, len(specs),
            self._terms(specs), _lib.ptr(records), _lib.ptr(ghosts), int(bool(has_lo)), int(bool(has_hi)),
            _lib.ptr(out)))

    def apply_multi(self, specs, n_loc, ld, beta, out, steps=None):
        """y = beta*y + sum over specs (tri, matrix index, x): every term reads a
        slab of its own, no ghost time steps (stk_kron_pack_apply_multi_steps); the
        dictionary form only (`explicit` plans keep the plain form).  `steps`: per
        term None or (t_begin, t_end), the local time steps the term's time factor
        reads its input at (time_factor_steps of the host copy of the factor): the
        term then gets lanes for those steps only."""
        assert not self.explicit and 2 <= len(specs) <= 3
        terms = self._terms([(tri, k) for tri, k, _ in specs])
        xs = (ctypes.c_void_p * len(specs))(*[_lib.ptr(x) for _, _, x in specs])
        t0 = t1 = None
        if steps is not None and any(s_ is not None for s_ in steps):
            rng = [(0, n_loc) if s_ is None else s_ for s_ in steps]
            t0 = (ctypes.c_int32 * len(specs))(*[int(a) for a, _ in rng])
            t1 = (ctypes.c_int32 * len(specs))(*[int(b) for _, b in rng])
        _lib.check(_lib.lib().stk_kron_pack_apply_multi_steps(
            _lib.stream(), ctypes.byref(self.pattern), n_loc, ld, len(specs),
            terms, xs, t0, t1, beta, _lib.ptr(out)))

    def apply(self, specs, x, ghosts, n_loc, ld, beta, out):
        """y = beta*y + sum over specs (tri, matrix index) applied to x;
        `ghosts`: (M, 2) interleaved ghost time steps or None."""
        terms = self._terms(specs)
        _lib.check(_lib.lib().stk_kron_pack_apply(
            _lib.stream(), ctypes.byref(self._pattern_for(specs)), n_loc, ld, len(specs),
            terms, _lib.ptr(x), _lib.ptr(ghosts), beta, _lib.ptr(out)))



# ----------------------------------------------------------------------------
# Serial operators of the reference (flat NumPy vectors in and out).
# ----------------------------------------------------------------------------
_self_distributions = {}


def self_distribution(N, M):
    """The one-rank DofDistributionMPI of N time steps by M space dofs that the
    serial operators' device vectors live on (one object per shape: vectors of one
    shape add up)."""
    from .comm import Comm
    from .mpi_vector import DofDistributionMPI
    key = (int(N), int(M))
    if key not in _self_distributions:
        _self_distributions[key] = DofDistributionMPI(Comm(distributed=False), *key)
    return _self_distributions[key]


def device_vector(x, n_time):
    """A flat host vector of the serial operators (time-major: entry t * M + i, the
    ordering of np.kron, reference linop.py:9-13) as a device-resident vector: a
    KronVectorMPI on one rank.  DeviceLinearOperators map it to another one;
    host_vector brings it back."""
    from .mpi_vector import KronVectorMPI
    X = np.ascontiguousarray(np.asarray(x, dtype=np.float64)).reshape(n_time, -1)
    vec = KronVectorMPI(self_distribution(n_time, X.shape[1]))
    vec.scatter(X)
    return vec


def host_vector(vec):
    """The flat host copy of a device vector."""
    out = np.empty(vec.N * vec.M)
    vec.gather(out)
    return out


def _is_device_vector(x):
    from .mpi_vector import KronVectorMPI
    return isinstance(x, KronVectorMPI)


class DeviceLinearOperator(LinearOperator):
    """A serial operator with the reference's LinearOperator surface (flat NumPy
    vector in, flat NumPy vector out: reference linop.py:6-65 builds those with
    SciPy) that ALSO maps device vectors to device vectors.  `matvec` is the
    function of flat host vectors; `vec_apply` the one of device vectors (default:
    `matvec` itself -- a function written with `@` and `+` over operators of this
    class is the same for both).  `A + B` and `A @ B` of two operators of this class
    are operators of this class; with anything else SciPy's rules apply."""
    def __init__(self, shape, matvec, vec_apply=None):
        super().__init__(dtype=np.float64, shape=tuple(int(k) for k in shape))
        self._host_fn = matvec
        self._vec_fn = matvec if vec_apply is None else vec_apply

    def _matvec(self, x):
        return np.asarray(self._host_fn(np.asarray(x, dtype=np.float64).reshape(-1))).reshape(-1)

    def dot(self, x):
        if _is_device_vector(x):
            assert x.N * x.M == self.shape[1], 'dimension mismatch'
            out = self._vec_fn(x)
            assert out.N * out.M == self.shape[0]
            return out
        if isinstance(x, DeviceLinearOperator):
            assert self.shape[1] == x.shape[0], 'dimension mismatch'
            return DeviceLinearOperator((self.shape[0], x.shape[1]), lambda v: self.dot(x.dot(v)))
        return super().dot(x)

    # SciPy routes `*`, `@` and calls through dot()
    def __add__(self, x):
        if isinstance(x, DeviceLinearOperator):
            assert self.shape == x.shape, 'dimension mismatch'
            return DeviceLinearOperator(self.shape, lambda v: self.dot(v) + x.dot(v))
        return super().__add__(x)


def KronLinOp(mat_time, mat_space):
    """x -> (A kron B) x on the device (reference linop.py:6-15)."""
    from .mpi_kron import SerialKron
    N, K = mat_time.shape
    M, L = mat_space.shape
    op = SerialKron(mat_time, mat_space)
    return DeviceLinearOperator((N * M, K * L), op.matvec, op.apply_vec)


def BlockDiagLinOp(linops):
    """Block diagonal of space operators (reference linop.py:29-44).  Blocks
    that are the same operator object (the preconditioner repeats one block per
    wavelet level, heateq.py:81-85) are applied together, one right-hand side
    per column of a single slab."""
    linops = [as_space_op(op) for op in linops]
    height = sum(op.shape[0] for op in linops)
    width = sum(op.shape[1] for op in linops)
    shapes = {op.shape for op in linops}
    uniform = len(shapes) == 1 and next(iter(shapes))[0] == next(iter(shapes))[1]
    groups = {}
    for t, op in enumerate(linops):
        groups.setdefault(id(op), (op, []))[1].append(t)

    def matvec(x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if uniform:
            X = x.reshape(len(linops), -1)
            Y = np.empty_like(X)
            for op, ts in groups.values():
                Y[ts] = (op @ np.ascontiguousarray(X[ts].T)).T
            return Y.reshape(-1)
        y = np.zeros(height)
        start_r = start_c = 0
        for op in linops:
            end_r, end_c = start_r + op.shape[0], start_c + op.shape[1]
            y[start_r:end_r] += op @ x[start_c:end_c]
            start_r, start_c = end_r, end_c
        return y

    held = {}

    def vec_apply(vec):
        # device vectors: the blocks are the time steps of the slab -- the parallel
        # operator of the same name on one rank (mpi_kron.BlockDiagMPI: equal blocks in
        # one batched call)
        assert uniform, 'device vectors need square blocks of one size'
        assert vec.N == len(linops)
        if 'op' not in held:
            from .mpi_kron import BlockDiagMPI
            held['op'] = BlockDiagMPI(self_distribution(len(linops), next(iter(shapes))[0]), linops)
        return held['op'] @ vec

    return DeviceLinearOperator((height, width), matvec, vec_apply)


def BlockLinOp(linops):
    """A rectangular grid of operators as one operator: row i of `linops`
    produces the i-th slice of the output from the slices of the input that its
    blocks multiply (reference linop.py:47-65; unused by the drivers).  Blocks
    may be space operators, matrices or LinearOperators."""
    rows = [[as_space_op(b) if sp.issparse(b) or isinstance(b, np.ndarray)
             else b for b in row] for row in linops]
    heights = [row[0].shape[0] for row in rows]
    widths = [b.shape[1] for b in rows[0]]

    def matvec(x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        col_edges = np.concatenate([[0], np.cumsum(widths)])
        pieces = []
        for row, h in zip(rows, heights):
            acc = np.zeros(h)
            for b, lo, hi in zip(row, col_edges[:-1], col_edges[1:]):
                acc += b @ x[lo:hi]
            pieces.append(acc)
        return np.concatenate(pieces)

    return LinearOperator(matvec=matvec, shape=(sum(heights), sum(widths)))
