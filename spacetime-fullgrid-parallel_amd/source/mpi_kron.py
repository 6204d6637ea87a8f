"""Time-parallel space-time operators on the GPU (counterpart of reference
source/mpi_kron.py; same class names, constructor arguments and
``_matvec(vec_in, vec_out)`` / ``op @ vec`` protocol).

What differs from the reference is where the arithmetic runs: each ``_matvec``
enqueues hand-written HIP kernels of libstk on the slab in HBM.
``TridiagKronMatMPI`` with a CSR space factor is ONE kernel (time stencil and
CSR gather fused), and ``SumMPI`` over such terms fuses up to 4 of them in one
pass over x and y (``stk_kron_sum_apply``) instead of a temporary and a ``+=``
per term (reference mpi_kron.py:77-90).
"""
import ctypes

import numpy as np
import scipy.sparse
import torch

from . import _lib
from .comm import MPI
from .linop import (SpaceMatrix, SpaceOp, as_space_op, permute_rows,
                    row_order_for, union_pattern)
from .mpi_vector import DofDistributionMPI, KronVectorMPI


def as_matrix(operator):
    """Dense matrix of anything supporting ``@`` on NumPy blocks
    (reference mpi_kron.py:8-10)."""
    cols = operator.shape[1]
    return operator @ np.eye(cols)


class LinearOperatorMPI:
    """Base class for linear space-time operators parallelized in time
    (reference mpi_kron.py:13-59)."""

    # The reference times every apply with MPI.Wtime.  Kernels are
    # asynchronous; set this to True (the drivers do) and every apply is timed
    # on the device with a pair of HIP events, so that time_applies is the time
    # the GPU spent on it.
    sync_timing = False

    def __init__(self, dofs_distr):
        self.dofs_distr = dofs_distr
        self.N = dofs_distr.N
        self.M = dofs_distr.M
        self.num_applies = 0
        self.time_applies = 0
        self.time_communication = 0

    def __matmul__(self, x):
        assert isinstance(x, KronVectorMPI)
        if not LinearOperatorMPI.sync_timing:
            start_time = MPI.Wtime()
            y = self._matvec(x, x._like())
            self.num_applies += 1
            self.time_applies += MPI.Wtime() - start_time  # enqueue time only
            return y
        # device time of the apply: HIP events on the stream the kernels are
        # launched on, bracketing everything _matvec enqueues (and the gaps in
        # which the host waits for a halo)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        y = self._matvec(x, x._like())
        e1.record()
        e1.synchronize()
        self.num_applies += 1
        self.time_applies += e0.elapsed_time(e1) * 1e-3
        return y

    def time_per_apply(self):
        assert (self.time_applies)
        return (self.time_applies / self.num_applies,
                self.time_communication / self.num_applies)

    def as_global_matrix(self):
        """Applies the operator to every unit vector (reference
        mpi_kron.py:38-59).  Expensive: tests only."""
        n = self.N * self.M
        I = np.eye(n)
        rank = self.dofs_distr.comm.Get_rank()
        result = x_glob = None
        if rank == 0:
            x_glob = np.empty(n)
            result = np.zeros((n, n))
        for k in range(n):
            x_mpi = KronVectorMPI(self.dofs_distr)
            x_mpi.scatter(I[k, :] if rank == 0 else None)
            x_mpi = self @ x_mpi
            x_mpi.gather(x_glob)
            if rank == 0:
                result[:, k] = x_glob
        return result


class IdentityMPI(LinearOperatorMPI):
    def __init__(self, dofs_distr):
        super().__init__(dofs_distr)

    def _matvec(self, vec_in, vec_out):
        vec_out.buf.copy_(vec_in.buf)
        return vec_out


def _local_tridiag(dofs_distr, mat_time):
    """(3, n_loc) coefficients [sub | diag | super] of the local rows of a
    tridiagonal time matrix (the slicing of reference mpi_kron.py:165-183)."""
    assert (scipy.sparse.isspmatrix_csr(mat_time))
    N, K = mat_time.shape
    assert (N == K)
    coo = mat_time.tocoo()
    assert np.all(np.abs(coo.row - coo.col) <= 1), 'time matrix not tridiagonal'
    tb, te = dofs_distr.t_begin, dofs_distr.t_end
    tri = np.zeros((3, te - tb))
    for r, c, v in zip(coo.row, coo.col, coo.data):
        if tb <= r < te:
            tri[c - r + 1, r - tb] += v
    return tri


class SumMPI(LinearOperatorMPI):
    """sum_k L_k (reference mpi_kron.py:71-90).  Consecutive
    TridiagKronMatMPI terms whose space factor is a plain matrix are fused
    into one kernel launch per group of 3 (ELL kernel) or 4 (CSR kernel)."""
    def __init__(self, dofs_distr, linops):
        assert all(isinstance(linop, LinearOperatorMPI) for linop in linops)
        self.linops = linops
        super().__init__(dofs_distr)
        self._groups = self._plan()

    def _plan(self):
        groups, run = [], []
        for op in self.linops:
            if isinstance(op, TridiagKronMatMPI) and op.fusable:
                run.append(op)
                if len(run) == _FusedKronSum.max_terms():
                    groups.append(_FusedKronSum(self.dofs_distr, run))
                    run = []
            else:
                if run:
                    groups.append(_FusedKronSum(self.dofs_distr, run))
                    run = []
                groups.append(op)
        if run:
            groups.append(_FusedKronSum(self.dofs_distr, run))
        return groups

    def _matvec(self, vec_in, vec_out):
        assert (vec_in is not vec_out)
        self.time_communication = 0
        first = True
        vec_tmp = None
        for g in self._groups:
            if isinstance(g, _FusedKronSum):
                self.time_communication += g.apply(vec_in, vec_out,
                                                   beta=0.0 if first else 1.0)
            else:
                if first:
                    g._matvec(vec_in, vec_out)
                else:
                    if vec_tmp is None:
                        vec_tmp = vec_in._like()
                    g._matvec(vec_in, vec_tmp)
                    vec_out += vec_tmp
                self.time_communication += g.time_communication
            first = False
        vec_out.communicated_bdr = False
        return vec_out


class CompositeMPI(LinearOperatorMPI):
    """L_1 L_2 ... x, right to left (reference mpi_kron.py:93-110)."""
    def __init__(self, dofs_distr, linops):
        assert all(isinstance(linop, LinearOperatorMPI) for linop in linops)
        N, M = linops[0].N, linops[0].M
        assert all(linop.N == N and linop.M == M for linop in linops)
        self.linops = linops
        super().__init__(dofs_distr)

    def _matvec(self, vec_in, vec_out):
        assert (vec_in is not vec_out)
        self.time_communication = 0
        Y = vec_in
        for linop in reversed(self.linops):
            Y = linop @ Y
            self.time_communication += linop.time_communication
        # hand the last result's storage over instead of copying it
        vec_out._buf = Y.buf
        vec_out.communicated_bdr = False
        return vec_out


_SIDE = []


def _side_stream():
    """The second HIP stream of the process (independent halves of an operator)."""
    if not _SIDE:
        _SIDE.append(torch.cuda.Stream())
    return _SIDE[0]


class BlockDiagMPI(LinearOperatorMPI):
    """y[t] = C_t x[t] for a list of space operators indexed by the GLOBAL
    time index (reference mpi_kron.py:113-132).

    Equal operator objects are applied to all their time slices in one batched
    call.  When every block is ``CompositeLinOp([C, A, C])`` with the C's taken
    from one MultiGrid family (multigrid.MultiGridFamily), all slices run
    through a single batched V-cycle with per-slice matrix coefficients.

    two_streams (default False): two column ranges of the slab through C A C side
    by side on two HIP streams, the way S runs its two independent K applies
    (heateq_mpi.SchurMPI).  Bit-identical, and measured SLOWER for P -- 21.1 ->
    25.9 ms on 65-step slabs, 5.5 -> 10.0 ms on 9-step slabs
    (profiles/r03_op_times_two_streams_P_*.log): the two halves read the same
    cache lines of every row and each streams the matrices once more.  The K
    applies of S gain because they are whole, independent slabs."""
    two_streams = False
    # (EllMatrices plan, index of the middle matrix in it) or None: the caller may
    # name the packed plan that holds the middle factor of its C A C blocks
    # (heateq_mpi.HeatEquationMPI does: A_x is matrix 1 of the (M_x, A_x) plan)
    # MEASURED and left off (profiles/r04_c_op_J6_J9_*.log, r04_b_launches_by_grid_J3_J9.txt):
    # the one-term packed pass is SLOWER than the row engine's own 5-slot copy of A_x,
    # 0.374 against 0.301 ms on 65-step slabs (row pairs: 10 gathers per two rows, half
    # of them for columns only M_x has), 0.085 against 0.058 ms on 9-step slabs.
    mid_packed = None
    pack_mid = False

    def __init__(self, dofs_distr, matrices_space):
        M = matrices_space[0].shape[0]
        for mat in matrices_space:
            assert mat.shape == (M, M)
        self.matrices_space = [as_space_op(m) for m in matrices_space]
        super().__init__(dofs_distr)
        self._local = self.matrices_space[dofs_distr.t_begin:dofs_distr.t_end]
        self._batched = self._try_batch()
        self._groups = None  # general case: (operator, slice count, slice indices on the device)

    def _try_batch(self):
        from .linop import CompositeLinOp
        from .multigrid import MultiGrid
        ops = self._local
        if all(op is ops[0] for op in ops):
            return ('uniform', ops[0])
        fam = None
        mids = set()
        for op in ops:
            if not (isinstance(op, CompositeLinOp) and len(op.linops) == 3
                    and op.linops[0] is op.linops[2]
                    and isinstance(op.linops[0], MultiGrid)
                    and op.linops[0].family is not None):
                return None
            if fam is None:
                fam = op.linops[0].family
            if op.linops[0].family is not fam:
                return None
            mids.add(id(op.linops[1]))
        if len(mids) != 1:
            return None
        members = [op.linops[0].member for op in ops]
        return ('family', fam, fam.slice_tables(members), ops[0].linops[1])

    def _matvec(self, vec_in, vec_out):
        assert (isinstance(vec_in, KronVectorMPI))
        assert (self.N == vec_in.N and self.M == vec_in.M)
        assert (vec_in.buf.shape == vec_out.buf.shape)
        assert (vec_out is not vec_in)
        n_loc = vec_in.n_loc
        b = self._batched
        if b is not None and b[0] == 'uniform':
            b[1].apply(vec_in.buf, out=vec_out.buf, n_loc=n_loc)
        elif b is not None and b[0] == 'family':
            _, fam, (cm, kind), mid = b
            half = ((n_loc + 1) // 2 + 1) & ~1  # even: the second range starts on a 16-byte pair
            if type(self).two_streams and n_loc >= 8 and vec_in.buf.is_cuda and hasattr(mid, 'mat'):
                # time slices are independent: two column ranges of the slab go
                # through C A C side by side on two HIP streams (a twin plan owns
                # the second set of level workspaces), filling each other's launch
                # gaps and tails like the two K applies inside S
                x, ld = vec_in.buf, vec_in.ld
                t1, t2 = torch.empty_like(x), torch.empty_like(x)
                main = torch.cuda.current_stream()
                side = _side_stream()
                side.wait_stream(main)
                for (off, n), stream, twin in (((0, half), main, False),
                                               ((half, n_loc - half), side, True)):
                    with torch.cuda.stream(stream):
                        cols = lambda t: t[:, off:]
                        kw = dict(n_loc=n, cm=cm[off:], kind=kind[off:], twin=twin, ld=ld)
                        fam.apply(cols(x), out=cols(t1), **kw)
                        mid.apply(cols(t1), out=cols(t2), n_loc=n, ld=ld)
                        fam.apply(cols(t2), out=cols(vec_out.buf), **kw)
                for t in (x, t1, t2, vec_out.buf):
                    t.record_stream(side)
                main.wait_stream(side)
            else:
                t1 = fam.apply(vec_in.buf, n_loc=n_loc, cm=cm, kind=kind)
                packed = None
                if self.mid_packed is not None and type(self).pack_mid:
                    packed = self.mid_packed[0].packed_for(n_loc)
                if packed is not None and packed.ok and packed.rows_per_unit == 2:
                    # I kron A_x on the packed slot stream S and the metric's operator
                    # share (4 bytes per slot instead of 12), where that stream serves
                    # row PAIRS (5 gathers per row, as many as A_x's own rows have; the
                    # one-row form gathers the union pattern's 7: 0.085 against 0.058 ms
                    # on 9-step slabs, profiles/r04_b_launches_by_grid_J3_J9.txt)
                    t2 = torch.empty_like(t1)
                    packed.apply([(None, self.mid_packed[1])], t1, None, n_loc,
                                 vec_in.ld, 0.0, t2)
                else:
                    t2 = mid.apply(t1, n_loc=n_loc)
                fam.apply(t2, out=vec_out.buf, n_loc=n_loc, cm=cm, kind=kind)
        else:
            # general case: the time slices of every distinct operator object
            # together (one slice at a time if all operators differ)
            vec_out.buf.zero_()
            if self._groups is None:
                groups = {}
                for t_loc, linop in enumerate(self._local):
                    groups.setdefault(id(linop), (linop, []))[1].append(t_loc)
                self._groups = [(linop, len(cols), _lib.to_dev(np.asarray(cols, dtype=np.int32)))
                                for linop, cols in groups.values()]
            lib, M, ld = _lib.lib(), vec_in.M, vec_in.ld
            for linop, n_cols, cols in self._groups:
                width = n_cols + (n_cols & 1)
                xin = torch.empty((M, width), dtype=torch.float64, device=vec_in.buf.device)
                _lib.check(lib.stk_slab_gather_columns(
                    _lib.stream(), M, n_cols, _lib.ptr(cols), _lib.ptr(vec_in.buf), ld,
                    _lib.ptr(xin), width))
                res = linop.apply(xin, n_loc=n_cols)
                _lib.check(lib.stk_slab_scatter_columns(
                    _lib.stream(), M, n_cols, _lib.ptr(cols), _lib.ptr(res), res.shape[1],
                    _lib.ptr(vec_out.buf), ld))
        vec_out.communicated_bdr = False
        return vec_out


class IdentityKronMatMPI(LinearOperatorMPI):
    """I_t kron M_x (reference mpi_kron.py:135-150)."""
    def __init__(self, dofs_distr, mat_space):
        M, L = mat_space.shape
        assert (M == L)
        self.mat_space = mat_space
        self.space_op = as_space_op(mat_space)
        super().__init__(dofs_distr)

    def _matvec(self, vec_in, vec_out):
        assert (isinstance(vec_in, KronVectorMPI))
        assert (self.N == vec_in.N and self.M == vec_in.M)
        assert (vec_in.buf.shape == vec_out.buf.shape)
        if vec_in is vec_out:  # allowed by the reference (mpi_kron.py:216)
            vec_out._buf = self.space_op.apply(vec_in.buf, n_loc=vec_in.n_loc)
        else:
            self.space_op.apply(vec_in.buf, out=vec_out.buf,
                                n_loc=vec_in.n_loc)
        vec_out.communicated_bdr = False
        return vec_out


class _TimeCSR:
    """Local rows of a sparse time matrix on the device; columns are local
    time indices, or n_loc + slot for rows that live on other ranks."""
    def __init__(self, n_loc, rows, cols, vals):
        m = scipy.sparse.csr_matrix((vals, (rows, cols)),
                                    shape=(n_loc, max(cols, default=0) + 1))
        m.sort_indices()
        self.indptr = _lib.to_dev(m.indptr.astype(np.int32))
        # a slab may own no entry at all (e.g. no node of a coarse wavelet
        # level): keep the arrays non-empty so their pointers are valid
        cols_ = m.indices.astype(np.int32) if m.nnz else np.zeros(1, np.int32)
        vals_ = m.data.astype(np.float64) if m.nnz else np.zeros(1)
        self.cols = _lib.to_dev(cols_)
        self.vals = _lib.to_dev(vals_)

    def apply(self, vec_in, recv, add_identity, vec_out):
        _lib.check(_lib.lib().stk_time_csr_apply(
            _lib.stream(), vec_in.M, vec_in.n_loc, vec_in.ld,
            _lib.ptr(self.indptr), _lib.ptr(self.cols), _lib.ptr(self.vals),
            _lib.ptr(vec_in.buf), _lib.ptr(recv), int(add_identity),
            _lib.ptr(vec_out.buf)))


class TridiagKronIdentityMPI(LinearOperatorMPI):
    """T_t kron I_x for a tridiagonal T_t: one ghost time row from each
    neighbour rank (reference mpi_kron.py:153-201)."""
    def __init__(self, dofs_distr, mat_time):
        self.tri = _local_tridiag(dofs_distr, mat_time)
        super().__init__(dofs_distr)
        n_loc = dofs_distr.t_end - dofs_distr.t_begin
        rows, cols, vals = [], [], []
        for t in range(n_loc):
            for d in (0, 1, 2):
                v = self.tri[d, t]
                if v == 0.0:
                    continue
                c = t + d - 1
                if c < 0:
                    c = n_loc  # ghost slot 0 = X_lo
                elif c >= n_loc:
                    c = n_loc + 1  # ghost slot 1 = X_hi
                rows.append(t), cols.append(c), vals.append(v)
        self._csr = _TimeCSR(n_loc, rows, cols, vals)

    def _matvec(self, vec_in, vec_out):
        assert (isinstance(vec_in, KronVectorMPI))
        assert (self.N == vec_in.N and self.M == vec_in.M)
        assert (vec_in.buf.shape == vec_out.buf.shape)
        assert (vec_in is not vec_out)
        self.time_communication += vec_in.communicate_bdr()
        ghost = None
        if vec_in.X_lo is not None or vec_in.X_hi is not None:
            ghost = vec_in.ghost_pair()
        self._csr.apply(vec_in, ghost, False, vec_out)
        vec_out.communicated_bdr = False
        return vec_out


class TridiagKronMatMPI(LinearOperatorMPI):
    """T_t kron M_x (reference mpi_kron.py:204-222).  One fused kernel when
    M_x is a plain matrix; otherwise the time factor, then the space operator
    on the result, as the reference does."""
    def __init__(self, dofs_distr, mat_time, mat_space):
        super().__init__(dofs_distr)
        self.mat_time = mat_time
        self.mat_space = mat_space
        self.space_op = as_space_op(mat_space)
        self.fusable = isinstance(self.space_op, SpaceMatrix)
        if self.fusable:
            self._fused = _FusedKronSum(dofs_distr, [self])
        else:
            self.I_M = IdentityKronMatMPI(dofs_distr, self.space_op)
            self.T_I = TridiagKronIdentityMPI(dofs_distr, mat_time)

    def _matvec(self, vec_in, vec_out):
        if self.fusable:
            assert (vec_in is not vec_out)
            self.time_communication = self._fused.apply(vec_in, vec_out, 0.0)
        else:
            self.T_I._matvec(vec_in, vec_out)
            self.I_M._matvec(vec_out, vec_out)
            self.time_communication = (self.I_M.time_communication +
                                       self.T_I.time_communication)
        vec_out.communicated_bdr = False
        return vec_out

    def as_matrix(self):
        return np.kron(as_matrix(self.mat_time), as_matrix(self.mat_space))


class _FusedKronSum:
    """y = beta*y + sum_k (T_k kron X_k) x for up to 4 TridiagKronMatMPI terms
    with plain CSR space factors: shared pattern, one launch.  use_ell selects
    the persistent sliced-ELL kernel (default) or the plain CSR one."""
    use_ell = True
    use_pack = True  # packed matrix stream when the plan fits
    # Several ranks, packed form: True = the pass over the slab runs WITHOUT the
    # ghost steps while the halo exchange is in flight and a one-lane-per-row
    # kernel recomputes the first and last local step afterwards (the reference
    # overlaps the exchange with the interior rows, mpi_kron.py:193-200); False =
    # wait for the halo, then one pass with the ghost steps as an extra lane per
    # row.  A halo that is already there (cached) always takes the one-pass form.
    # Both forms, and the one-rank kernel, round every entry the same way.
    # Round 6: the two boundary steps are RECOMPUTED (what makes the forms bit-equal).
    # From the slab itself that gathers 16 bytes per 128-byte line and costs 0.069 / 0.130
    # / 0.079 ms at 9 / 17 / 33 steps of 1 046 529 rows where adding a share cost 0.042 /
    # 0.052 / 0.041 ms; from the compact records the pack kernel leaves beside the rows it
    # extracts (stk_halo_pack_records, stk_kron_pack_boundary_apply) 0.038 / 0.038 / 0.035
    # ms: the overlapped form wins on every slab length again
    # (profiles/r06_slab_shapes_J9.log, _J10.log).
    overlap = True
    OVERLAP_FROM = 1

    @classmethod
    def max_terms(cls):
        return 3 if cls.use_ell else 4

    def __init__(self, dofs_distr, ops):
        assert 1 <= len(ops) <= self.max_terms()
        self.use_ell = type(self).use_ell
        self.dofs_distr = dofs_distr
        mats = [op.space_op.mat for op in ops]
        hints = [op.mat_space for op in ops]
        self.nnz_terms = [int(m.nnz) for m in mats]
        tris = [_local_tridiag(dofs_distr, op.mat_time) for op in ops]
        self.needs_lo = any(t[0, 0] != 0.0 for t in tris)
        self.needs_hi = any(t[2, -1] != 0.0 for t in tris)
        self.tri = [_lib.to_dev(t) for t in tris]
        self.n_terms = len(ops)
        if self.use_ell:
            from .linop import EllMatrices
            self.ell = EllMatrices.shared(mats, hints)
            self.row_ids = self.ell.row_ids
        else:
            indptr, indices, vals = union_pattern(mats)
            order = row_order_for(hints + mats, indptr, indices)
            indptr, indices, vals, row_ids = permute_rows(
                indptr, indices, vals, order)
            self.indptr = _lib.to_dev(indptr)
            self.indices = _lib.to_dev(indices)
            self.row_ids = None if row_ids is None else _lib.to_dev(row_ids)
            self.vals = [_lib.to_dev(v) for v in vals]
            self.terms = (_lib.KronTerm * len(ops))()

    def apply(self, vec_in, vec_out, beta=0.0):
        time_comm = 0.0
        packed = (self.ell.packed_for(vec_in.n_loc)
                  if self.use_ell and type(self).use_pack else None)
        if packed is not None and packed.ok:
            # one pass: matrix stream packed, ghost time steps handled by an extra
            # lane per row (csrc/kron_pack.hip); the halo has to be there first
            ghosts = None
            specs = [(self.tri[k], k) for k in range(self.n_terms)]
            halo = self.dofs_distr.size > 1 and (self.needs_lo or self.needs_hi)
            if (halo and type(self).overlap and not vec_in.communicated_bdr
                    and beta == 0.0 and vec_in.n_loc >= type(self).OVERLAP_FROM):
                # (the boundary steps are REWRITTEN afterwards: a beta != 0 would
                # need the old values the pass has replaced -- one-pass form then)
                time_comm = vec_in.communicate_bdr(callback=lambda: packed.apply(
                    specs, vec_in.buf, None, vec_in.n_loc, vec_in.ld, 0.0,
                    vec_out.buf), records=True)
                # the two boundary steps from the compact records the pack left and the
                # interleaved received rows: both sides in one lane per slot row
                packed.apply_boundary(specs, vec_in.boundary_records(), vec_in.ghost_interleaved(),
                                      self.needs_lo and vec_in.X_lo is not None,
                                      self.needs_hi and vec_in.X_hi is not None,
                                      vec_in.n_loc, vec_in.ld, vec_out.buf)
                return time_comm
            if halo:
                time_comm = vec_in.communicate_bdr()
                ghosts = vec_in.ghost_interleaved()
            packed.apply(specs, vec_in.buf, ghosts, vec_in.n_loc, vec_in.ld, beta,
                         vec_out.buf)
            return time_comm
        if self.use_ell:
            # the slab-local part runs while the halo exchange is in flight
            # (the reference overlaps the interior rows, mpi_kron.py:193-196)
            def local():
                self.ell.apply_local(
                    [(self.tri[k], k, vec_in.buf, None, None)
                     for k in range(self.n_terms)], vec_in.n_loc, vec_in.ld,
                    beta, vec_out.buf)

            if self.dofs_distr.size > 1 and (self.needs_lo or self.needs_hi):
                if beta != 0.0:
                    # the boundary steps are rewritten after the local part, which
                    # needs the old y they scale: the one-call form keeps a copy
                    time_comm = vec_in.communicate_bdr()
                    lo = vec_in.X_lo if self.needs_lo else None
                    hi = vec_in.X_hi if self.needs_hi else None
                    self.ell.apply([(self.tri[k], k, vec_in.buf, lo, hi)
                                    for k in range(self.n_terms)], vec_in.n_loc,
                                   vec_in.ld, beta, vec_out.buf)
                    return time_comm
                time_comm = vec_in.communicate_bdr(callback=local)
                lo = vec_in.X_lo if self.needs_lo else None
                hi = vec_in.X_hi if self.needs_hi else None
                self.ell.apply_ghost(
                    [(self.tri[k], k, vec_in.buf, lo, hi)
                     for k in range(self.n_terms)], vec_in.n_loc, vec_in.ld,
                    vec_out.buf)
            else:
                local()
            return time_comm
        if self.dofs_distr.size > 1:
            time_comm = vec_in.communicate_bdr()
        lo = vec_in.X_lo if self.needs_lo else None
        hi = vec_in.X_hi if self.needs_hi else None
        x = _lib.ptr(vec_in.buf)
        for k in range(self.n_terms):
            t = self.terms[k]
            t.tri, t.vals = _lib.ptr(self.tri[k]), _lib.ptr(self.vals[k])
            t.x, t.x_lo, t.x_hi = x, _lib.ptr(lo), _lib.ptr(hi)
        _lib.check(_lib.lib().stk_kron_sum_apply(
            _lib.stream(), vec_in.M, vec_in.n_loc, vec_in.ld,
            _lib.ptr(self.indptr), _lib.ptr(self.indices),
            _lib.ptr(self.row_ids), self.n_terms, self.terms, beta,
            _lib.ptr(vec_out.buf)))
        return time_comm

    def phase_times(self, vec_in, vec_out, reps=10):
        """Device milliseconds of the pieces of one multi-rank apply on the packed
        path, each timed alone with HIP events on the halo that is already there:
        the pack of the two boundary rows, the pass over the slab without the ghost
        steps (what runs beside the exchange), the ghost steps' share afterwards, and
        the one-pass form with ghost lanes.  None without the packed path or a GPU."""
        packed = (self.ell.packed_for(vec_in.n_loc)
                  if self.use_ell and type(self).use_pack else None)
        if packed is None or not packed.ok or not vec_in.buf.is_cuda:
            return None
        if self.dofs_distr.size > 1:
            vec_in.communicate_bdr()
        n_loc, ld, M = vec_in.n_loc, vec_in.ld, vec_in.M
        specs = [(self.tri[k], k) for k in range(self.n_terms)]
        lo = vec_in.X_lo if self.needs_lo else None
        hi = vec_in.X_hi if self.needs_hi else None
        send = torch.empty((2, M), dtype=torch.float64, device=vec_in.buf.device)

        def timed(fn):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        records = torch.empty((M, 4), dtype=torch.float64, device=vec_in.buf.device)
        out = {'pack_ms': timed(lambda: _lib.check(_lib.lib().stk_halo_pack_records(
            _lib.stream(), M, n_loc, ld, _lib.ptr(vec_in.buf), _lib.ptr(send[0]), 1,
            _lib.ptr(send[1]), 1, _lib.ptr(records)))),
            'pass_without_ghosts_ms': timed(lambda: packed.apply(
                specs, vec_in.buf, None, n_loc, ld, 0.0, vec_out.buf))}
        if lo is not None or hi is not None:
            ghosts = vec_in.ghost_interleaved()
            # (the key keeps its round-5 name: the boundary steps once the halo is there)
            out['ghost_share_ms'] = timed(lambda: packed.apply_boundary(
                specs, records, ghosts, lo is not None, hi is not None, n_loc, ld, vec_out.buf))
            out['one_pass_with_ghost_lanes_ms'] = timed(lambda: packed.apply(
                specs, vec_in.buf, ghosts, n_loc, ld, 0.0, vec_out.buf))
        return out

    def kernel_name(self, n_loc):
        """Name of the kernel instantiation `apply` launches (for the bench
        line and for matching a PMC record to the build)."""
        if not self.use_ell:
            return 'kron_sum_kernel<%d>' % self.n_terms
        if type(self).use_pack and self.ell.packed_for(n_loc).ok:
            ghost = self.dofs_distr.size > 1 and (self.needs_lo or self.needs_hi)
            pk = self.ell.packed_for(n_loc)
            overlapped = ghost and type(self).overlap and n_loc >= type(self).OVERLAP_FROM
            return 'kron_pack_kernel<%d, %d, %s, %s%s>' % (
                self.n_terms, pk.K,
                'pass without ghost steps beside the exchange + boundary kernel' if overlapped
                else 'ghost lanes' if ghost else 'no ghosts',
                'row pairs' if pk.rows_per_unit == 2 else 'single rows',
                ', explicit values' if pk.explicit else '')
        return 'kron_ell_kernel<%d, shared input, %d>' % (self.n_terms, self.ell.K)

    def algorithmic_bytes(self, n_loc, M):
        """Bytes one apply must move: x once, y once,
        ghost rows, every CSR array once."""
        h = int(self.needs_lo and self.dofs_distr.rank > 0) + int(
            self.needs_hi and self.dofs_distr.rank + 1 < self.dofs_distr.size)
        return (16 * n_loc * M + 8 * h * M + 12 * sum(self.nnz_terms) +
                4 * (M + 1) * len(self.nnz_terms))


class SparseKronIdentityMPI(LinearOperatorMPI):
    """M_t kron I_x for a sparse time matrix with symmetric sparsity pattern;
    rows of other ranks are fetched point-to-point
    (reference mpi_kron.py:259-317)."""
    def __init__(self, dofs_distr, mat_time, add_identity=False):
        super().__init__(dofs_distr)
        assert scipy.sparse.isspmatrix_csr(mat_time)
        N, K = mat_time.shape
        assert (N == K)
        assert (mat_time.nnz)
        self.add_identity = add_identity
        tb, te = dofs_distr.t_begin, dofs_distr.t_end
        coo = mat_time.tocoo()
        keep = (coo.row >= tb) & (coo.row < te)
        self.row, self.col, self.data = (coo.row[keep], coo.col[keep],
                                         coo.data[keep])
        self.comm_dofs = sorted(
            set((int(r), int(c)) for r, c in zip(self.row, self.col)
                if c < tb or te <= c))
        need = sorted(set(c for _, c in self.comm_dofs))
        slot = {c: k for k, c in enumerate(need)}
        n_loc = te - tb
        cols = [
            int(c - tb) if tb <= c < te else n_loc + slot[int(c)]
            for c in self.col
        ]
        self._csr = _TimeCSR(n_loc, list(self.row - tb), cols,
                             list(self.data))

    def _matvec(self, vec_in, vec_out):
        assert (isinstance(vec_in, KronVectorMPI))
        assert (self.N == vec_in.N and self.M == vec_in.M)
        assert (vec_in.buf.shape == vec_out.buf.shape)
        assert vec_out is not vec_in
        recv = None
        if len(self.comm_dofs):
            recv, _, reqs = vec_in.communicate_dofs(self.comm_dofs)
            start_time = MPI.Wtime()
            self.dofs_distr.comm.wait_all(reqs)
            self.time_communication += MPI.Wtime() - start_time
        self._csr.apply(vec_in, recv, self.add_identity, vec_out)
        vec_out.communicated_bdr = False
        return vec_out


class MatKronIdentityMPI(LinearOperatorMPI):
    """M_t kron I_x for a general (dense) time matrix through an all-to-all
    transpose of the vector (reference mpi_kron.py:225-256)."""
    single_rank_shortcut = True  # False: the transposes also on one rank (tests)

    def __init__(self, dofs_distr, mat_time):
        N, K = mat_time.shape
        assert (N == K)
        self.mat_time = mat_time
        if hasattr(mat_time, 'levels'):
            self.levels = mat_time.levels
        super().__init__(dofs_distr)
        dense = mat_time if isinstance(mat_time, np.ndarray) else (
            mat_time.toarray() if scipy.sparse.issparse(mat_time) else
            as_matrix(mat_time))
        self._time_op = SpaceMatrix(scipy.sparse.csr_matrix(dense))
        self._dense = np.ascontiguousarray(dense, dtype=np.float64)
        self._local_csr = None
        # the one-rank shortcut below runs one thread per output over the entries of its
        # row: good for rows of a few entries (W: 1.69 against 2.46 ms through the
        # transposes at config 3), bad when some rows are full (W^T: 5.6 against 2.6 ms)
        self._max_row_nnz = int((self._dense != 0).sum(axis=1).max()) if self._dense.size else 0

    def _matvec(self, vec_in, vec_out):
        assert (isinstance(vec_in, KronVectorMPI))
        assert (self.N == vec_in.N and self.M == vec_in.M)
        assert (vec_in.buf.shape == vec_out.buf.shape)
        if (self.dofs_distr.size == 1 and vec_in.buf.is_cuda and type(self).single_rank_shortcut
                and self._max_row_nnz <= 16):
            # one rank holds every time row of every space dof already: the transposes
            # of the reference (mpi_kron.py:246-253) have nothing to exchange, the time
            # factor acts on the contiguous time column of each dof -- through its
            # non-zero entries (stk_time_csr_apply; the wavelet matrix has N log N of N^2)
            if self._local_csr is None:
                rows, cols = np.nonzero(self._dense)
                self._local_csr = _TimeCSR(self.N, list(rows), list(cols),
                                           list(self._dense[rows, cols]))
            self._local_csr.apply(vec_in, None, False, vec_out)
            vec_out.communicated_bdr = False
            return vec_out
        vec_perm, comm_time = vec_in.permute()
        self.time_communication += comm_time
        # the permuted vector has the time index as its "space" index
        vec_perm._buf = self._time_op.apply(vec_perm.buf,
                                            n_loc=vec_perm.n_loc)
        _, comm_time = vec_perm.permute(vec_out)
        self.time_communication += comm_time
        vec_out.communicated_bdr = False
        return vec_out


class SerialKron:
    """(A kron B) on a flat host vector, run on the device (single rank);
    backs linop.KronLinOp (reference linop.py:6-15).  The time factor may be
    rectangular and may be a matrix or any LinearOperator (it is small: it is
    applied as a dense matrix); the space factor is a matrix or a space
    operator (multigrid, a direct inverse, a composite)."""
    def __init__(self, mat_time, mat_space):
        self.N, self.K = mat_time.shape
        self.M, self.L = mat_space.shape
        if scipy.sparse.issparse(mat_time):
            dense = mat_time.toarray()
        elif isinstance(mat_time, np.ndarray):
            dense = mat_time
        else:  # a LinearOperator, e.g. WaveletTransformOp or its transpose
            dense = mat_time @ np.eye(self.K)
        self._time = _lib.to_dev(np.ascontiguousarray(dense, dtype=np.float64))
        self._space = as_space_op(mat_space)

    def _on_slab(self, xin, ld_in):
        """(L, ld_in) slab of K time steps -> (M, ld_out) slab of N."""
        ld_out = self.N + (self.N & 1)
        z = torch.empty((self.L, ld_out), dtype=torch.float64, device=xin.device)
        _lib.check(_lib.lib().stk_time_dense_apply(
            _lib.stream(), self.L, self.K, ld_in, self.N, ld_out,
            _lib.ptr(self._time), _lib.ptr(xin), _lib.ptr(z)))
        return self._space.apply(z, n_loc=self.N)

    def apply_vec(self, vec):
        """The same map on a device vector (linop.device_vector: a one-rank
        KronVectorMPI of K time steps by L space dofs): nothing leaves the device."""
        from .linop import self_distribution
        assert isinstance(vec, KronVectorMPI) and vec.dofs_distr.size == 1
        assert (vec.N, vec.M) == (self.K, self.L), 'dimension mismatch'
        y = self._on_slab(vec.buf, vec.ld)
        return KronVectorMPI.around(self_distribution(self.N, self.M), y)

    def matvec(self, x):
        X = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(self.K, self.L))
        ld_in, ld_out = self.K + (self.K & 1), self.N + (self.N & 1)
        dev = _lib.compute_device()
        lib, st = _lib.lib(), _lib.stream()
        # the reference's time-major host block <-> the space-major slab: libstk's
        # upload / download (transposed on the device, padding columns zero)
        xin = torch.empty((self.L, ld_in), dtype=torch.float64, device=dev)
        _lib.check(lib.stk_slab_upload(st, self.L, self.K, ld_in, X.ctypes.data,
                                       _lib.ptr(xin)))
        z = torch.empty((self.L, ld_out), dtype=torch.float64, device=dev)
        _lib.check(lib.stk_time_dense_apply(
            st, self.L, self.K, ld_in, self.N, ld_out,
            _lib.ptr(self._time), _lib.ptr(xin), _lib.ptr(z)))
        y = self._space.apply(z, n_loc=self.N)
        out = np.empty((self.N, self.M))
        _lib.check(lib.stk_slab_download(st, self.M, self.N, y.shape[1], _lib.ptr(y),
                                         out.ctypes.data))
        return out.reshape(-1)
