"""Spatial multigrid with Gauss-Seidel smoothing on the GPU (counterpart of
reference source/multigrid.py).

``MeshHierarchy`` and ``MultiGrid`` keep the reference's constructor
signatures.  Setup (Galerkin products, reference multigrid.py:142-145) is done
once on the host with SciPy, as in the reference; every V-cycle then runs in
libstk (``stk_mg_apply``) on all time slices of the slab at once.

The smoother is the reference's sequential Gauss-Seidel sweep in dof order
(multigrid.py:83-97, which is what it asks of PETSc MatSOR, :116-127).  It is
run in parallel without changing its result by scheduling rows along the
dependency DAG of the sweep; see csrc/mg.hip and mesh.py.

``MultiGridFamily`` is an addition: the preconditioner needs one multigrid per
wavelet level for the matrices 2^j M_x + alpha A_x (reference
heateq_mpi.py:97-98, 147-153).  A family stores A and M hierarchies once and
applies all members in one batched V-cycle with a per-time-slice coefficient.
"""
import ctypes
import threading

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from .assembly import prolongation_matrices
from .linop import (EllRowsMatrix, SpaceOp, tile_order_from_coords,
                    tile_rows_from_coords,
                    union_pattern)


class MeshHierarchy:
    """Prolongation / restriction matrices between the free dofs of the levels
    of a hierarchically numbered mesh (reference multigrid.py:14-80).  Takes a
    mesh.TriangleMesh (the reference takes an NGSolve H1 space), or a ready
    list of prolongation matrices."""
    def __init__(self, fes=None, shared_comm=None, P_mats=None):
        self.coords = None
        if P_mats is None:
            mesh = getattr(fes, 'mesh', fes)
            P_mats = prolongation_matrices(mesh)
            # coordinates of the free dofs (level l = the first n_l of them):
            # only used to pick cache-friendly row orders on the device
            self.coords = mesh.points[~mesh.boundary]
        self.shared_comm = shared_comm
        self.P_mats = [sp.csr_matrix(P) for P in P_mats]
        self.R_mats = [P.T.tocsr() for P in self.P_mats]
        self.J = len(self.P_mats)
        # what several plans on this hierarchy need alike (K's and the preconditioner
        # family's are built side by side): computed once, by whoever asks first
        self._shared, self._shared_lock = {}, threading.Lock()

    def shared(self, key, make):
        """make() once per key for all plans built on this hierarchy (row orders,
        transfer operators on the device: they depend on the mesh alone)."""
        with self._shared_lock:
            slot = self._shared.get(key)
            if slot is None:
                slot = self._shared[key] = {'lock': threading.Lock()}
        with slot['lock']:  # a second asker waits for the first one's result
            if 'value' not in slot:
                slot['value'] = make()
        return slot['value']

    def forget(self):
        """Drops what depends on matrix VALUES -- Galerkin chains, R A products and the
        device copies of R, P they were formed from -- once the plans that asked for
        them are built (the plans keep what they point into).  Those entries are keyed
        by the ARRAYS of the matrices (_mat_key): the matrices handed to MultiGrid /
        MultiGridFamily are treated as immutable while plans are being built on a
        hierarchy, and a caller that changes values in place calls this before it
        builds the next plan (HeatEquationMPI calls it at the end of its set-up).  What
        depends on the mesh alone (tile orders, bands, transfer copies) stays."""
        with self._shared_lock:
            for key in [k for k in self._shared if k[0] in ('galerkin', 'galerkin_rp', 'ra')]:
                del self._shared[key]

    def tile_order(self, n):
        """Mesh-tile processing order of the first n dofs (index order without
        coordinates)."""
        if self.coords is None:
            return np.arange(n, dtype=np.int32)
        return self.shared(('tile', n), lambda: tile_order_from_coords(self.coords[:n]))

    def coord_band(self, n):
        """Slices of the first n dofs along the last coordinate axis (mesh rows of a
        structured mesh): the candidate bands of coupling_bands."""
        if self.coords is None:
            return None

        def make():
            _, band = np.unique(np.round(np.asarray(self.coords)[:n, -1], 12), return_inverse=True)
            return band.astype(np.int64)

        return self.shared(('band', n), make)

    def transfer_copies(self, j):
        """P and R between levels j - 1 and j on the device, as CSR and as ELL copies in
        tile order: they depend on the mesh alone, one set serves every plan of the
        hierarchy (read-only in all kernels)."""
        def make():
            P = sp.csr_matrix(self.P_mats[j - 1])
            R = sp.csr_matrix(self.R_mats[j - 1])
            dev, ells = {}, {}
            for name, m, order in (('p', P, self.tile_order(P.shape[0])),
                                   ('r', R, self.tile_order(P.shape[1]))):
                m.sort_indices()
                dev[name + '_indptr'] = _lib.to_dev(np.asarray(m.indptr, dtype=np.int32))
                dev[name + '_indices'] = _lib.to_dev(np.asarray(m.indices, dtype=np.int32))
                dev[name + '_vals'] = _lib.to_dev(np.asarray(m.data, dtype=np.float64))
                ells[name] = EllRowsMatrix(m.indptr, m.indices, m.data, None, order)
            return dev, ells

        return self.shared(('transfer', j), make)

    def prepare(self):
        """Everything plans share, ahead of time (the driver calls it in the thread
        that builds the hierarchy, beside the assembly of the matrices)."""
        sizes = [P.shape[1] for P in self.P_mats] + ([self.P_mats[-1].shape[0]] if self.P_mats else [])
        # finest first: the plans start with their finest level, and it is the long one
        for n in reversed(sizes):
            self.tile_order(n)
            self.coord_band(n)
        if _lib.compute_device().type == 'cuda':
            for j in reversed(range(1, self.J + 1)):
                self.transfer_copies(j)
        return self


def _tile_order(hierarchy, n):
    """Processing order of the first n dofs of a hierarchy (any object with P_mats /
    R_mats / coords; a MeshHierarchy computes every order once for all its plans)."""
    if hasattr(hierarchy, 'tile_order'):
        return hierarchy.tile_order(n)
    coords = getattr(hierarchy, 'coords', None)
    return np.arange(n, dtype=np.int32) if coords is None else tile_order_from_coords(coords[:n])


def _shared(hierarchy, key, make):
    return hierarchy.shared(key, make) if hasattr(hierarchy, 'shared') else make()


def _mat_key(mat):
    """Identity of a CSR matrix by its arrays (sp.csr_matrix(m) wraps, it does not copy)."""
    return (mat.indptr.ctypes.data, mat.indices.ctypes.data, mat.data.ctypes.data, mat.nnz)


def _drop_roundoff(mat, rel=1e-14):
    """Remove stored entries of a Galerkin product that are pure rounding noise
    (|a_ij| < rel * max|a|).  On nested right-triangle meshes R A P is the coarse
    stiffness matrix, whose hypotenuse couplings are exactly 0; SciPy keeps them
    as ~1e-17 entries, which would cost the sweep a dependency level and two ELL
    slots for contributions below the rounding error of the other terms."""
    if mat.nnz:
        mat.data[np.abs(mat.data) < rel * np.abs(mat.data).max()] = 0.0
        mat.eliminate_zeros()
    return mat


# Experiments only: callable(fine matrix, Galerkin chain coarse -> fine) -> chain, applied to
# every chain a plan builds (which coarse levels own a deviation: replace them and look).
CHAIN_HOOK = None


def galerkin_product(R, A, P, cache=None):
    """R A P (reference multigrid.py:142-145) as a SciPy CSR matrix with sorted
    rows.  On a GPU the product is formed by libstk (stk_csr_galerkin: one coarse
    row per thread, every sum accumulated in the order and with the roundings of
    SciPy's `(R @ A) @ P`, so the result is bit for bit the host's -- asserted by
    test_device_plan_construction_matches_host); without one, or for rows longer
    than the kernel holds, by SciPy itself.  `cache`: dict that keeps the device
    copies of R and P between the products of one level."""
    R, A = sp.csr_matrix(R), sp.csr_matrix(A)
    P = None if P is None else sp.csr_matrix(P)  # None: R A alone (restricted-residual product)

    def on_host():
        out = sp.csr_matrix(R @ A if P is None else R @ A @ P)
        out.sort_indices()
        return out

    if _lib.compute_device().type != 'cuda' or R.shape[0] < 64:
        return on_host()

    def upload(m):
        return (_lib.to_dev(np.asarray(m.indptr, dtype=np.int32)), _lib.to_dev(np.asarray(m.indices, dtype=np.int32)),
                _lib.to_dev(np.asarray(m.data, dtype=np.float64)))

    def up(m, key=None):
        if cache is None or key is None:
            return upload(m)
        # the per-level dict is shared by the plan builders' threads (K's chain and the
        # family's two run side by side): filled under its own lock
        with cache.setdefault('lock', threading.Lock()):
            if key not in cache:
                cache[key] = upload(m)
            return cache[key]

    nc, cap = R.shape[0], (16 if P is not None else 32)
    dR, dA = up(R, 'R'), up(A)  # `cache` is per level: the role names the matrix
    dP = up(P, 'P') if P is not None else (None, None, None)
    dev = dR[0].device
    counts = torch.empty(nc, dtype=torch.int32, device=dev)
    idx = torch.empty((nc, cap), dtype=torch.int32, device=dev)
    val = torch.empty((nc, cap), dtype=torch.float64, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().stk_csr_galerkin(
        _lib.stream(), nc, *[_lib.ptr(t) for t in dR + dA + dP], cap, _lib.ptr(counts),
        _lib.ptr(idx), _lib.ptr(val), _lib.ptr(overflow)))
    if int(overflow.item()) != 0:  # a row with more than `cap` entries: SciPy does it
        return on_host()
    counts = counts.cpu().numpy().astype(np.int64)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    mask = np.arange(cap)[None, :] < counts[:, None]
    indptr = np.concatenate([[0], np.cumsum(counts)])
    out = sp.csr_matrix((val[mask], idx[mask], indptr),
                        shape=(nc, A.shape[1] if P is None else P.shape[1]))
    out.has_sorted_indices = True
    return out


def _dev_csr(m):
    return (_lib.to_dev(np.asarray(m.indptr, dtype=np.int32)), _lib.to_dev(np.asarray(m.indices, dtype=np.int32)),
            _lib.to_dev(np.asarray(m.data, dtype=np.float64)))


def member_chains(hierarchy, mat_a, mat_m, ca, cms, keep_rows=8192):
    """Galerkin chains (reference multigrid.py:142-145) of the ASSEMBLED matrices
    cm_k M + ca A, k = 0 .. K-1 -- what the reference builds one MultiGrid per wavelet
    level from (heateq_mpi.py:97-98, 147-153) -- formed on the device, values only:
    the fine matrices differ in their values alone (one union pattern), every product
    R C P is stk_csr_galerkin's (SciPy's order and roundings: bit for bit `R @ C @ P`,
    as galerkin_product), the rounding noise is dropped as _drop_roundoff does, and
    nothing but the small levels ever reaches the host.

    A preconditioner family (MultiGridFamily) combines TWO chains per time slice,
    ca (R A P) + cm (R M P).  In exact arithmetic that is the chain above; in floating
    point the assembled chain carries ITS roundings from level to level, and what
    separates the two grows fourfold per level down -- 2e-16 relative on the first
    coarse level, 1e-12 on level 0 of a ten-level hierarchy (a rounding residue that is
    the same in every stencil acts like a multiple of the mass matrix, which scales
    with h^2).  On the smallest levels that difference owns most of the gap between the
    family's r.Pr history and the reference's
    (profiles/r06_history_by_coarse_level_J7_J10.json), so the family hands the
    reference's matrices to the coarse end of its plan (stk_mg_set_member_matrices).

    Returns chains[k][level] = SciPy CSR for the levels of at most `keep_rows` rows,
    or None where a product has rows longer than the kernel holds (the caller keeps
    the combination)."""
    mat_a, mat_m = sp.csr_matrix(mat_a), sp.csr_matrix(mat_m)
    scaled = sp.csr_matrix(float(ca) * mat_a)  # fl(ca a), as in `cm * M + ca * A`
    indptr, indices, (va, vm) = union_pattern([scaled, mat_m])
    J = hierarchy.J
    d_ptr, d_idx = _lib.to_dev(np.asarray(indptr, dtype=np.int32)), _lib.to_dev(np.asarray(indices, dtype=np.int32))
    d_va, d_vm = _lib.to_dev(np.asarray(va, dtype=np.float64)), _lib.to_dev(np.asarray(vm, dtype=np.float64))
    dev = d_ptr.device
    cap = 16
    lanes = torch.arange(cap, device=dev, dtype=torch.int32)[None, :]

    def host_csr(ptr, idx, data, shape):
        out = sp.csr_matrix((data.cpu().numpy(), idx.cpu().numpy(), ptr.cpu().numpy()), shape=shape)
        out.has_sorted_indices = True
        return out

    chains = []
    for cm in cms:
        # SciPy: fl(fl(cm m) + fl(ca a)) on the union pattern (entries of one matrix alone: copied)
        data = d_vm * float(cm) + d_va
        ptr, idx, n_cols = d_ptr, d_idx, mat_a.shape[1]
        kept = {}
        for j in reversed(range(J)):
            R, P = sp.csr_matrix(hierarchy.R_mats[j]), sp.csr_matrix(hierarchy.P_mats[j])
            nc = R.shape[0]
            if nc < 64:  # galerkin_product forms these on the host: so here
                C = host_csr(ptr, idx, data, (R.shape[1], n_cols))
                for jj in reversed(range(j + 1)):
                    Rj, Pj = sp.csr_matrix(hierarchy.R_mats[jj]), sp.csr_matrix(hierarchy.P_mats[jj])
                    C = sp.csr_matrix(Rj @ C @ Pj)
                    C.sort_indices()
                    kept[jj] = C = _drop_roundoff(C)
                break
            held = _shared(hierarchy, ('galerkin_rp', j), dict)
            with held.setdefault('lock', threading.Lock()):
                for key, m in (('R', R), ('P', P)):
                    if key not in held:
                        held[key] = _dev_csr(m)
            dR, dP = held['R'], held['P']
            counts = torch.empty(nc, dtype=torch.int32, device=dev)
            oidx = torch.empty((nc, cap), dtype=torch.int32, device=dev)
            oval = torch.empty((nc, cap), dtype=torch.float64, device=dev)
            overflow = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(_lib.lib().stk_csr_galerkin(
                _lib.stream(), nc, *[_lib.ptr(t) for t in dR + (ptr, idx, data) + dP], cap, _lib.ptr(counts),
                _lib.ptr(oidx), _lib.ptr(oval), _lib.ptr(overflow)))
            if int(overflow.item()) != 0:
                return None
            valid = lanes < counts[:, None]
            mag = oval.abs()
            biggest = torch.where(valid, mag, torch.zeros_like(mag)).max()
            keep = valid & (mag >= 1e-14 * biggest) & (oval != 0.0)  # _drop_roundoff + eliminate_zeros
            ptr = torch.zeros(nc + 1, dtype=torch.int32, device=dev)
            ptr[1:] = torch.cumsum(keep.sum(dim=1), dim=0).to(torch.int32)
            idx, data, n_cols = oidx[keep].contiguous(), oval[keep].contiguous(), P.shape[1]
            if nc <= keep_rows:
                kept[j] = host_csr(ptr, idx, data, (nc, n_cols))
        chains.append(kept)
    return chains


def _depth_on_device(n, d_indptr, d_indices, backward):
    """The DAG depth of every row by repeated relaxation on the device
    (stk_gs_depth_step) -- the same fixed point as the NumPy loop below."""
    dev = d_indptr.device
    depth = [torch.zeros(n, dtype=torch.int32, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev)]
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(n + 1):
        changed.zero_()
        _lib.check(_lib.lib().stk_gs_depth_step(
            _lib.stream(), n, _lib.ptr(d_indptr), _lib.ptr(d_indices), int(backward),
            _lib.ptr(depth[it & 1]), _lib.ptr(depth[1 - (it & 1)]), _lib.ptr(changed)))
        if int(changed.item()) == 0:
            break
    return depth[it & 1].cpu().numpy().astype(np.int64)


def gauss_seidel_schedule(indptr, indices, backward=False, on_device=None):
    """Groups the rows of a CSR pattern by their depth in the dependency DAG
    of a Gauss-Seidel sweep in dof order: row i must wait for its neighbours
    j < i (forward) or j > i (backward).  Returns (ptr, rows): rows of group g
    are rows[ptr[g]:ptr[g+1]], ascending (descending for backward), and are
    mutually independent.  `on_device`: (indptr, indices) of the pattern as
    device tensors -- the depths are then computed by libstk."""
    n = len(indptr) - 1
    if on_device is not None and n > 0 and on_device[0].is_cuda:
        depth = _depth_on_device(n, on_device[0], on_device[1], backward)
        return _groups_by_depth(depth, n, backward)
    rows_of = np.repeat(np.arange(n), np.diff(indptr))
    dep = indices > rows_of if backward else indices < rows_of
    depth = np.zeros(n, dtype=np.int64)
    nonempty = np.flatnonzero(np.diff(indptr) > 0)
    starts = indptr[:-1][nonempty]
    while True:
        cand = np.where(dep, depth[indices] + 1, 0)
        new = np.zeros(n, dtype=np.int64)
        if len(cand):
            new[nonempty] = np.maximum.reduceat(cand, starts)
        if np.array_equal(new, depth):
            break
        depth = new
    return _groups_by_depth(depth, n, backward)


def _groups_by_depth(depth, n, backward):
    order = np.argsort(depth, kind='stable')
    if backward:
        # descending row index inside a group, like the sequential sweep
        order = np.lexsort((-np.arange(n), depth))
    counts = np.bincount(depth, minlength=int(depth.max()) + 1 if n else 1)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return ptr, order.astype(np.int32)


def coupling_bands(coords, indptr, indices, rows_of=None, coord_band=None):
    """A band index per row such that two coupled rows always lie in the same
    or in adjacent bands -- the property the strip-wise Gauss-Seidel sweep of
    csrc/mg.hip rests on -- with bands as thin as that allows.  The property is
    VERIFIED on the pattern, never assumed.  With coordinates the bands are
    slices along the last axis (mesh rows of a structured mesh, halved until the
    property holds on an unstructured one); without, the levels of a
    breadth-first search, which have it by construction for a symmetric
    pattern.  Returns None when there is no useful banding."""
    n = len(indptr) - 1
    if n < 2:
        return None
    if rows_of is None:
        rows_of = np.repeat(np.arange(n), np.diff(indptr))

    def ok(b):
        return len(indices) == 0 or np.abs(b[rows_of] - b[indices]).max() <= 1

    verified = False
    if coords is not None:
        if coord_band is not None:  # the slices along the last axis, computed by the caller
            band = coord_band.copy()
        else:
            _, band = np.unique(np.round(np.asarray(coords)[:n, -1], 12),
                                return_inverse=True)
            band = band.astype(np.int64)
        while band.max() > 0:
            if ok(band):
                verified = True
                break
            band //= 2
    else:
        from scipy.sparse.csgraph import dijkstra
        pat = sp.csr_matrix((np.ones(len(indices)), indices, indptr),
                            shape=(n, n))
        dist = dijkstra(pat + pat.T, unweighted=True, indices=0)
        if not np.isfinite(dist).all():
            return None
        band = dist.astype(np.int64)
    if band.max() < 1 or not (verified or ok(band)):
        return None
    return band


# Mesh rows per band of the strip-wise sweeps (constructor argument `band_merge` of
# MultiGrid / MultiGridFamily; None = these defaults).  coupling_bands gives the thinnest
# bands the coupling allows (single mesh rows on a structured mesh); the rows of a
# dependency group are listed band by band and in tile order inside a band.  With one
# mesh row per band a stage walks the level's full width before it comes back to the row
# above; with several mesh rows per band the tile order inside the (thicker) band keeps
# the two readers of a gathered row within a tile's width.  Coupled rows still lie at most
# one band apart (coarser bands keep the property, which mg.hip's strips rest on) and the
# rows of a launch are independent: results do not change by a bit.  Measured
# (profiles/r05_band_merge.log): the preconditioner family's P 21.6 -> 21.1 ms at 65 steps
# and 26.0 -> 25.5 ms on config 5's 17-step slab with 4-8 rows per band, 5.36 -> 5.41 ms at
# 9 steps; K's plans (two chains side by side inside S) gain nothing.  HeatEquationMPI
# asks for 6 rows per band in the family's plan from 16 time steps on.
BAND_MERGE = int(__import__('os').environ.get('STK_BAND_MERGE', '1'))
BAND_MERGE_FAMILY = int(__import__('os').environ.get('STK_BAND_MERGE_FAMILY', str(BAND_MERGE)))

# Gauss-Seidel rows on the device: True = diagonal-free copies,
# u_i = (f_i - sum_{j != i} a_ij u_j) / a_ii (PETSc MatSOR's form; one gather less
# per row); False = the whole row in the slots and the reference's pure-Python
# update u_i += (f_i - row_i u) / a_ii (reference multigrid.py:89-97) -- part of
# the "reference arithmetic" mode (heateq_mpi.HeatEquationMPI(arithmetic='reference')).
GS_DIAG_FREE = True
# Levels whose Gauss-Seidel copies follow GS_DIAG_FREE when it is set; the others
# get the full rows.  None = every level; a callable (level, finest) -> bool.
GS_DIAG_FREE_LEVELS = None
# True: a level that gets the full rows also gets the diagonal-free copies as its
# ALTERNATIVE form (stk_mg_level.ell_fwd_alt / ell_bwd_alt): the plan option
# "fast_until_cycle" then runs the first V-cycles of an application on them.
GS_ALT_COPIES = False


class _DeviceHierarchy:
    """Everything one libstk multigrid plan needs, resident on the device."""
    def __init__(self, mat_a, mat_m, hierarchy, smoothsteps, vcycles,
                 coarse_mats, gs_rows=None, band_merge=None, member_mats=None):
        # row form of the Gauss-Seidel copies: 'free' (diagonal-free on every level),
        # 'full' (the reference's form on every level), 'owned' (the reference's form
        # on the finest level, which also gets the diagonal-free copies as its
        # alternative form; diagonal-free below -- what HeatEquationMPI's default
        # arithmetic runs on), None: the module switches GS_DIAG_FREE* above
        assert gs_rows in (None, 'free', 'full', 'owned')
        self.gs_rows = gs_rows
        if band_merge is None:
            band_merge = BAND_MERGE_FAMILY if mat_m is not None else BAND_MERGE
        self.band_merge = max(1, int(band_merge))
        self.J = hierarchy.J
        self.smoothsteps, self.vcycles = smoothsteps, vcycles
        self.has_m = mat_m is not None
        # Galerkin hierarchies, coarse to fine (reference multigrid.py:142-145).  The
        # chain of a matrix is the same whichever plan asks (K's and the family's both
        # start from A_x): formed once per hierarchy and matrix.
        def chain(fine):
            fine = sp.csr_matrix(fine)

            def make():
                mats = [fine]
                for j in reversed(range(self.J)):
                    R, P = hierarchy.R_mats[j], hierarchy.P_mats[j]
                    held = _shared(hierarchy, ('galerkin_rp', j), dict)  # device copies of R, P
                    mats.insert(0, _drop_roundoff(galerkin_product(R, mats[0], P, held)))
                if CHAIN_HOOK is not None:  # experiments: tools/history_by_coarse_level.py
                    mats = CHAIN_HOOK(fine, mats)
                return mats

            return list(_shared(hierarchy, ('galerkin', _mat_key(fine)), make))

        A = chain(mat_a)
        Mm = chain(mat_m) if self.has_m else None
        self.mats_a, self.mats_m = A, Mm
        self.shape = A[-1].shape
        self._keep = []  # device tensors / host arrays the plan points into
        self.levels = (_lib.MGLevel * (self.J + 1))()
        # the levels are independent host work (NumPy / SciPy release the GIL):
        # finest first, side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=3) as pool:
            for done in [pool.submit(_lib.in_device_context(self._fill_level), j, hierarchy)
                         for j in reversed(range(self.J + 1))]:
                done.result()
        inv = np.stack([np.linalg.inv(np.asarray(m.todense()))
                        for m in coarse_mats(A[0], Mm[0] if Mm else None)])
        self.coarse_inv = _lib.to_dev(np.ascontiguousarray(inv))
        self.n_kinds = inv.shape[0]
        # member_mats[k][level]: the level matrices of member k itself (member_chains) for
        # the coarse end of the plan; coarse-inverse kind k + 1 names member k (a callable:
        # the chains are formed beside this constructor and joined by coarse_mats above)
        if callable(member_mats):
            member_mats = member_mats()
        self.member_mats = member_mats
        self.member_levels = 0  # levels 1 .. member_levels of the plans run on them
        if member_mats is not None:
            # entries outside the plan's pattern would be lost: keep the combination then
            def pattern(m):
                m = sp.csr_matrix(m)
                return sp.csr_matrix((np.ones(m.nnz), m.indices, m.indptr), shape=m.shape)

            for kept in member_mats:
                for level, C in kept.items():
                    if 1 <= level < self.J:
                        have = pattern(abs(A[level]) + abs(Mm[level]))
                        outside = pattern(C) - pattern(C).multiply(have)
                        outside.eliminate_zeros()
                        if outside.nnz:
                            self.member_mats = None
        self.plan = None
        self.plan_ld = 0
        self.twin = None  # a second plan on the same matrices with workspaces of its own
        self.twin_ld = 0
        self.options = {}  # stk_mg_set_option keys of this hierarchy's plans

    def _row_form(self, j):
        """(rows with their diagonal?, diagonal-free copies beside them?) of level j."""
        if self.gs_rows == 'free':
            return False, False
        if self.gs_rows == 'full':
            return True, False
        if self.gs_rows == 'owned':
            return j == self.J, j == self.J
        free = GS_DIAG_FREE and (GS_DIAG_FREE_LEVELS is None
                                 or GS_DIAG_FREE_LEVELS(j, self.J))
        return not free, (not free) and GS_DIAG_FREE and GS_ALT_COPIES

    def _fill_level(self, j, hierarchy):
        L = self.levels[j]
        mats = [self.mats_a[j]] + ([self.mats_m[j]] if self.has_m else [])
        indptr, indices, vals = union_pattern(mats)
        n = len(indptr) - 1
        rows_of = np.repeat(np.arange(n), np.diff(indptr))
        diag = np.full(n, -1, dtype=np.int64)
        on_diag = np.flatnonzero(indices == rows_of)
        diag[rows_of[on_diag]] = on_diag
        assert (diag >= 0).all(), 'matrix lacks a diagonal entry'
        dev = {
            'indptr': _lib.to_dev(indptr),
            'indices': _lib.to_dev(indices),
            'vals_a': _lib.to_dev(vals[0]),
            'diag': _lib.to_dev(diag.astype(np.int32)),
        }
        if self.has_m:
            dev['vals_m'] = _lib.to_dev(vals[1])
        L.n = n
        host = {}
        if j > 0:
            # processing order: mesh tiles if coordinates are known
            tile = _tile_order(hierarchy, n)
            rank = np.empty(n, dtype=np.int64)
            rank[tile] = np.arange(n)
            # bands for the strip-wise sweeps: coupled rows at most one band apart
            band = coupling_bands(hierarchy.coords, indptr, indices, rows_of,
                                  hierarchy.coord_band(n) if hasattr(hierarchy, 'coord_band') else None)
            if band is not None and self.band_merge > 1 and hierarchy.coords is not None:
                band = band // self.band_merge
            key = rank if band is None else band * np.int64(n) + rank
            vm = vals[1] if self.has_m else None
            # transfer operators and the restricted-residual product R A: independent
            # of the sweeps' copies, built beside them
            moved = {}
            transfers = threading.Thread(
                target=_lib.in_device_context(self._transfer_ells),
                args=(j, hierarchy, mats, tile, moved))
            transfers.start()
            ells = {'a': EllRowsMatrix(indptr, indices, vals[0], vm, tile)}
            for name, bw in (('fwd', False), ('bwd', True)):
                ptr, rows = gauss_seidel_schedule(indptr, indices, bw,
                                                  on_device=(dev['indptr'], dev['indices']))
                host[name + '_ptr'] = ptr
                dev[name + '_rows'] = _lib.to_dev(rows)
                setattr(L, 'n_' + name, len(ptr) - 1)
                setattr(L, name + '_ptr_host',
                        ptr.ctypes.data_as(ctypes.c_void_p))
                # the same groups as one ELL matrix, each group band by band and
                # in tile order inside a band
                groups = [rows[ptr[g]:ptr[g + 1]] for g in range(len(ptr) - 1)]
                groups = [r_[np.argsort(key[r_], kind='stable')] for r_ in groups]
                listed = (np.concatenate(groups) if n else
                          np.zeros(0, dtype=np.int64))
                full_rows, alt_copies = self._row_form(j)
                if not full_rows:
                    ells[name] = EllRowsMatrix(indptr, indices, vals[0], vm,
                                               listed, diag=True)
                else:
                    ells[name] = EllRowsMatrix(indptr, indices, vals[0], vm, listed,
                                               dia_values=True)
                    if alt_copies:
                        ells[name + '_alt'] = EllRowsMatrix(indptr, indices, vals[0], vm,
                                                            listed, diag=True)
                if band is not None:
                    # band of every ELL position (ascending inside a group):
                    # lets the plan run the sweeps strip by strip (mg.hip)
                    host[name + '_trow'] = np.ascontiguousarray(
                        band[listed], dtype=np.int32)
                    setattr(L, name + '_tile_row_host',
                            host[name + '_trow'].ctypes.data_as(ctypes.c_void_p))
                    L.n_tile_rows = int(band.max()) + 1
                host[name + '_pos'] = ptr  # group g = positions ptr[g]:ptr[g+1]
                setattr(L, name + '_pos_host',
                        ptr.ctypes.data_as(ctypes.c_void_p))
                if not bw:
                    fwd_groups = groups
            transfers.join()
            if 'error' in moved:
                raise moved['error']
            dev.update(moved['dev'])
            ells.update(moved['ells'])
            if all(e.ok for e in ells.values()):
                if moved['ra'].ok:
                    ells['ra'] = moved['ra']
                for name, e in ells.items():
                    setattr(L, 'ell_' + name, ctypes.pointer(e.struct))
                host['ells'] = ells
                self._zero_start_ells(L, host, indptr, indices, vals[0], vm,
                                      diag, fwd_groups)
                # one read of the device-side overflow flags per level: a listed row
                # longer than its slots (a Gauss-Seidel copy of a matrix without a
                # diagonal entry) must not reach the sweeps
                EllRowsMatrix.check_all(list(ells.values()) + list(host.get('fwd0', ([],))[0]))
        for name, t in dev.items():
            setattr(L, name, _lib.ptr(t))
        self._keep.append((dev, host))
        if j == self.J:
            self.groups_fwd = len(host['fwd_ptr']) - 1 if j > 0 else 0

    def _transfer_ells(self, j, hierarchy, mats, tile, out):
        """P, R of level j (CSR on the device + ELL copies) and the product R A
        (restricted residual in one step: d = (R A) u - R f); results into `out`."""
        try:
            P = sp.csr_matrix(hierarchy.P_mats[j - 1])
            R = sp.csr_matrix(hierarchy.R_mats[j - 1])
            nc = P.shape[1]
            tile_c = _tile_order(hierarchy, nc)

            def transfer_copies():
                dev_, ells_ = {}, {}
                for name, m, order in (('p', P, tile), ('r', R, tile_c)):
                    m.sort_indices()
                    dev_[name + '_indptr'] = _lib.to_dev(np.asarray(m.indptr, dtype=np.int32))
                    dev_[name + '_indices'] = _lib.to_dev(np.asarray(m.indices, dtype=np.int32))
                    dev_[name + '_vals'] = _lib.to_dev(np.asarray(m.data, dtype=np.float64))
                    ells_[name] = EllRowsMatrix(m.indptr, m.indices, m.data, None, order)
                return dev_, ells_

            # P and R and their ELL copies depend on the mesh alone: one set on the
            # device for all plans of a MeshHierarchy
            if hasattr(hierarchy, 'transfer_copies'):
                shared_dev, shared_ells = hierarchy.transfer_copies(j)
            else:
                shared_dev, shared_ells = transfer_copies()
            dev, ells = dict(shared_dev), dict(shared_ells)
            prods = [_shared(hierarchy, ('ra', j, _mat_key(m)),
                             lambda m=m: _drop_roundoff(galerkin_product(R, m, None))) for m in mats]
            ra_ptr, ra_idx, ra_vals = union_pattern(prods)
            out['ra'] = EllRowsMatrix(ra_ptr, ra_idx, ra_vals[0],
                                      ra_vals[1] if self.has_m else None, tile_c)
            out['dev'], out['ells'] = dev, ells
        except Exception as exc:  # re-raised by the level's thread
            out['error'] = exc

    def _zero_start_ells(self, L, host, indptr, indices, va, vm, diag,
                         groups):
        """The first forward sweep of a level visit starts from u = 0
        (reference multigrid.py:176, 187): every product with a not yet
        updated neighbour is an exact zero.  Per dependency group, an ELL copy
        that keeps only the entries whose column lies in an EARLIER group --
        narrower rows, fewer gathers, and no need to zero u beforehand.  Unused
        slots repeat the row's first kept column (a row of an earlier group: written
        before this row in every order the sweeps run in, strip-wise included);
        group 0 keeps nothing, gathers from f instead of u (mg.hip) and pads with
        its own row."""
        n = len(indptr) - 1
        if not groups or any(len(g) == 0 for g in groups):
            return
        grp = np.empty(n, dtype=np.int64)
        for g, rows in enumerate(groups):
            grp[rows] = g
        rows_of = np.repeat(np.arange(n), np.diff(indptr))
        kept = np.bincount(rows_of[grp[indices] < grp[rows_of]], minlength=n)
        ells = [EllRowsMatrix(indptr, indices, va, vm, rows, pad_col=-1,
                              dia_values=True, earlier_group=grp, kept_counts=kept)
                for rows in groups]
        if not all(e.ok for e in ells):
            return
        arr = (_lib.EllRows * len(ells))(*[e.struct for e in ells])
        host['fwd0'] = (ells, arr)
        L.ell_fwd0 = ctypes.cast(arr, ctypes.POINTER(_lib.EllRows))

    def _create_plan(self, ld):
        handle = ctypes.c_void_p()
        _lib.check(_lib.lib().stk_mg_create(
            self.J + 1, self.levels, self.smoothsteps, self.vcycles,
            self.n_kinds, _lib.ptr(self.coarse_inv), ld,
            ctypes.byref(handle)))
        for key, value in self.options.items():
            _lib.check(_lib.lib().stk_mg_set_option(handle, key.encode(), int(value)))
        if self.member_mats is not None:
            self._hand_over_members(handle)
        return handle

    def _hand_over_members(self, handle):
        lib = _lib.lib()
        Lc = int(lib.stk_mg_coarse_levels(handle))
        if Lc < 1 or not all(l in kept for kept in self.member_mats for l in range(1, Lc + 1)):
            return
        n_kinds = self.n_kinds
        assert n_kinds == len(self.member_mats) + 1  # kind 0: A alone
        for level in range(1, Lc + 1):
            ptrs = [(ctypes.c_void_p * n_kinds)() for _ in range(3)]
            keep = []
            for k, kept in enumerate(self.member_mats):
                C = kept[level]
                arrs = (np.ascontiguousarray(C.indptr, dtype=np.int32), np.ascontiguousarray(C.indices, dtype=np.int32),
                        np.ascontiguousarray(C.data, dtype=np.float64))
                keep.append(arrs)
                for which in range(3):
                    ptrs[which][k + 1] = arrs[which].ctypes.data
            _lib.check(lib.stk_mg_set_member_matrices(handle, level, n_kinds, *ptrs))
        self.member_levels = Lc

    def ensure_plan(self, ld, twin=False):
        """The plan for slabs up to `ld` columns.  twin=True: a second plan on the
        SAME matrices with level workspaces of its own, so that two applies can be
        in flight at once on two streams (the two K applies inside S)."""
        name, name_ld = ('twin', 'twin_ld') if twin else ('plan', 'plan_ld')
        have = getattr(self, name)
        if have is not None and ld <= getattr(self, name_ld):
            return have
        if have is not None:
            _lib.check(_lib.lib().stk_mg_destroy(have))
        handle = self._create_plan(ld)
        setattr(self, name, handle)
        setattr(self, name_ld, ld)
        return handle

    def set_option(self, key, value):
        self.options[key] = int(value)
        for handle in (self.plan, self.twin):
            if handle is not None:
                _lib.check(_lib.lib().stk_mg_set_option(handle, key.encode(), int(value)))

    def apply(self, x, out, n_loc, ca, cm, kind, twin=False, ld=None):
        """`ld`: leading dimension when x / out are column ranges of wider slabs
        (their data pointers then start inside a row)."""
        ld = x.shape[1] if ld is None else ld
        n_loc = ld if n_loc is None else n_loc
        if out is None:
            out = torch.empty_like(x)
        plan = self.ensure_plan(ld, twin)
        _lib.check(_lib.lib().stk_mg_apply(plan, _lib.stream(), n_loc, ld, ca,
                                           _lib.ptr(cm), _lib.ptr(kind),
                                           _lib.ptr(x), _lib.ptr(out)))
        return out

    def __del__(self):
        try:
            for handle in (self.plan, self.twin):
                if handle is not None:
                    _lib.lib().stk_mg_destroy(handle)
        except Exception:
            pass


class MultiGrid(SpaceOp):
    """`vcycles` V-cycles from a zero initial guess, `smoothsteps` forward
    Gauss-Seidel sweeps before and backward sweeps after the coarse-grid
    correction, exact solve on level 0 (reference multigrid.py:130-197)."""
    family = None
    member = None

    def __init__(self, mat, hierarchy, smoothsteps=2, vcycles=1,
                 fuse_restrict=None, gs_rows=None, band_merge=None):
        self.num_applies = 0
        self.time_applies = 0
        self.hierarchy = hierarchy
        self.smoothsteps = smoothsteps
        self.vcycles = vcycles
        self._dev = _DeviceHierarchy(mat, None, hierarchy, smoothsteps,
                                     vcycles, lambda a0, m0: [a0], gs_rows=gs_rows,
                                     band_merge=band_merge)
        if fuse_restrict is not None:
            # False: the restricted residual as the reference forms it,
            # R (A u - f) (multigrid.py:174-175); see stk_mg_set_option
            self._dev.set_option('fuse_restrict', bool(fuse_restrict))
        self.mats = self._dev.mats_a
        self.shape = self.mats[-1].shape
        self.dtype = np.float64

    def apply(self, x, out=None, n_loc=None, twin=False, **kw):
        self.num_applies += 1
        return self._dev.apply(x, out, n_loc, 1.0, None, None, twin=twin)

    def _matvec(self, b):
        """The reference's entry point (multigrid.py:184-193): a NumPy vector in, `vcycles`
        V-cycles from zero, a NumPy vector out, wall time (device drained) added to
        time_applies.  The hot path calls apply() on device slabs and does not stop
        the clock for it."""
        import time
        torch.cuda.synchronize()
        began = time.perf_counter()
        x = SpaceOp.__matmul__(self, np.asarray(b, dtype=np.float64).reshape(-1))
        torch.cuda.synchronize()
        self.time_applies += time.perf_counter() - began
        return x

    matvec = _matvec

    def time_per_apply(self):
        """reference multigrid.py:195-197 (of the applies that went through _matvec)."""
        assert (self.time_applies)
        return self.time_applies / self.num_applies

    def apply_pair(self, x1, x2, n_loc=None, shared=()):
        """(K x1, K x2) with the two independent V-cycle chains side by side on two
        HIP streams (a twin plan owns the second set of level workspaces).  Two
        chains of ~300 dependent launches each fill each other's launch gaps and
        tails: measured 15.8 -> 14.9 ms on 65-step slabs, 3.66 -> 2.97 ms on 9-step
        slabs (profiles/r03_two_stream_k.log), bit-identical results.  x2 may be a
        callable that produces the second right-hand side; it then runs on the
        side stream too (`shared`: tensors of the main stream it reads)."""
        main = torch.cuda.current_stream()
        side = self._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if callable(x2):
                x2 = x2()
            y2 = self.apply(x2, n_loc=n_loc, twin=True)
        y1 = self.apply(x1, n_loc=n_loc)
        for t in (x2, y2) + tuple(t for t in shared if t is not None):
            t.record_stream(side)
            t.record_stream(main)
        main.wait_stream(side)
        return y1, y2

    _side = None

    def _side_stream(self):
        if MultiGrid._side is None:
            MultiGrid._side = torch.cuda.Stream()
        return MultiGrid._side

    def smooth(self, level, u, f, its, backward, n_loc=None):
        """`its` Gauss-Seidel sweeps on one level, in place on the slab u
        (PETScSMoother.PreSmooth / PostSmooth, reference multigrid.py:112-127)."""
        ld = u.shape[1]
        n_loc = ld if n_loc is None else n_loc
        plan = self._dev.ensure_plan(ld)
        _lib.check(_lib.lib().stk_mg_smooth(plan, _lib.stream(), level, n_loc,
                                            ld, 1.0, None, its, int(backward),
                                            _lib.ptr(f), _lib.ptr(u)))
        return u


class Smoother:
    """SOR smoother (reference multigrid.py:83-97): PreSmooth is one forward
    Gauss-Seidel sweep in dof order, u_i += (f_i - row_i u) / a_ii, PostSmooth one
    backward sweep; `u` is updated in place.  The reference loops over the rows in
    Python; here the rows are grouped by their depth in the sweep's dependency graph
    and a group is one launch of the row engine (csrc/rows_ell.hip, <GS>, rows with
    their diagonal) -- every row performs the sequential sweep's arithmetic on the
    same inputs (stk_mg_smooth).

    u, f: NumPy vectors of length n as in the reference (uploaded, swept, written
    back into u), NumPy arrays (n, k) -- k right-hand sides swept together -- or
    device slabs (n, ld), time fastest, as KronVectorMPI.buf holds them.
    `its`: sweeps per call (the reference's class does one)."""
    gs_rows = 'full'

    def __init__(self, mat, its=1):
        mat = sp.csr_matrix(mat)
        assert mat.shape[0] == mat.shape[1]
        self.its = int(its)
        self.shape = mat.shape
        # the sweeps of one matrix are the sweeps of the finest level of a two-level
        # plan whose coarse space is spanned by the first unknown (never visited here)
        n = mat.shape[0]
        P = sp.csr_matrix((np.ones(1), (np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64))),
                          shape=(n, 1))
        self._mg = MultiGrid(mat, MeshHierarchy(P_mats=[P]), smoothsteps=self.its, vcycles=1,
                             gs_rows=type(self).gs_rows)

    def _sweeps(self, u, f, backward):
        if torch.is_tensor(u):
            assert torch.is_tensor(f) and u.shape == f.shape and u.shape[0] == self.shape[0]
            self._mg.smooth(1, u, f, self.its, backward)
            return
        assert u.shape == f.shape and u.shape[0] == self.shape[0]
        cols = 1 if u.ndim == 1 else u.shape[1]
        ld = cols + (cols & 1)
        dev = []
        for a in (u, f):
            slab = torch.zeros((self.shape[0], ld), dtype=torch.float64, device=_lib.compute_device())
            slab[:, :cols] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, cols)).to(
                slab.device)
            dev.append(slab)
        self._mg.smooth(1, dev[0], dev[1], self.its, backward, n_loc=cols)
        u[...] = dev[0][:, :cols].cpu().numpy().reshape(u.shape)

    def PreSmooth(self, u, f):
        self._sweeps(u, f, False)

    def PostSmooth(self, u, f):
        self._sweeps(u, f, True)


class PETScSMoother(Smoother):
    """`its` forward (PreSmooth) or backward (PostSmooth) SOR sweeps, the smoother
    MultiGrid uses (reference multigrid.py:100-127, PETSc's MatSOR with omega = 1).
    MatSOR updates a row as u_i = (f_i - sum_{j != i} a_ij u_j) / a_ii; so do the
    diagonal-free Gauss-Seidel copies this class sweeps with (equal to Smoother's
    form up to rounding)."""
    gs_rows = 'free'

    def __init__(self, mat, its):
        super().__init__(mat, its)


class MultiGridFamily:
    """MultiGrid operators for the matrices ca * A + cm_k * M, k = 0..K-1,
    sharing one device hierarchy.  ``members[k]`` is a MultiGrid for matrix k;
    BlockDiagMPI recognises members of one family and runs all time slices in
    one batched V-cycle."""
    def __init__(self, mat_a, mat_m, hierarchy, ca, cms, smoothsteps=2,
                 vcycles=1, fuse_restrict=None, gs_rows=None, band_merge=None, exact_coarse=False):
        """exact_coarse: the coarse end of the plan (the levels of the fused coarse
        kernel, and the exact solve on level 0) runs on the Galerkin chain of every
        member's ASSEMBLED matrix cm_k M + ca A, as the reference's one-hierarchy-per-
        wavelet-level does, instead of the combination ca (R A P) + cm_k (R M P) of the
        two shared chains: see member_chains."""
        self.ca = float(ca)
        self.cms = [float(c) for c in cms]
        self.hierarchy = hierarchy
        # the members' chains are device work (Galerkin kernels, compactions) that needs
        # nothing of the plan: formed in a thread of their own beside the planner's host
        # work, joined where the coarsest matrices are inverted
        box = {}
        worker = None
        if exact_coarse and hierarchy.J >= 1 and _lib.compute_device().type == 'cuda':
            def run():
                try:
                    got = member_chains(hierarchy, mat_a, mat_m, self.ca, self.cms)
                    box['chains'] = got if got is not None and all(0 in kept for kept in got) else None
                except BaseException as err:  # re-raised by the planner's thread
                    box['error'] = err

            worker = threading.Thread(target=_lib.in_device_context(run), daemon=True)
            worker.start()

        def chains_now():
            if worker is not None:
                worker.join()
                if 'error' in box:
                    raise box['error']
            return box.get('chains')

        def coarse(a0, m0):
            # kind 0: A alone (unused by members); kind 1+k: member k
            chains = chains_now()
            if chains is not None:
                return [a0] + [kept[0] for kept in chains]
            return [a0] + [self.ca * a0 + c * m0 for c in self.cms]

        self._dev = _DeviceHierarchy(mat_a, mat_m, hierarchy, smoothsteps,
                                     vcycles, coarse, gs_rows=gs_rows, band_merge=band_merge,
                                     member_mats=chains_now)
        self.member_chains = chains_now()
        if fuse_restrict is not None:
            self._dev.set_option('fuse_restrict', bool(fuse_restrict))
        self.shape = self._dev.shape
        self.members = [_FamilyMember(self, k) for k in range(len(self.cms))]
        self._uniform = {}

    def slice_tables(self, members):
        """Per-time-slice coefficient and coarse-inverse index for a list of
        member indices (one per local time slice)."""
        cm = _lib.to_dev(np.array([self.cms[k] for k in members]))
        kind = _lib.to_dev(np.array([k + 1 for k in members], dtype=np.int32))
        return cm, kind

    def apply(self, x, out=None, n_loc=None, cm=None, kind=None, twin=False, ld=None):
        return self._dev.apply(x, out, n_loc, self.ca, cm, kind, twin=twin, ld=ld)

    def apply_member(self, k, x, out=None, n_loc=None):
        n = x.shape[1] if n_loc is None else n_loc
        key = (k, n)
        if key not in self._uniform:
            self._uniform[key] = self.slice_tables([k] * n)
        cm, kind = self._uniform[key]
        return self._dev.apply(x, out, n_loc, self.ca, cm, kind)


class _FamilyMember(MultiGrid):
    apply_pair = None  # members share the family's one plan: no second set of workspaces

    def __init__(self, family, k):
        self.family, self.member = family, k
        self.hierarchy = family.hierarchy
        self.shape = family.shape
        self.dtype = np.float64
        self.num_applies = 0
        self.time_applies = 0

    def apply(self, x, out=None, n_loc=None, **kw):
        self.num_applies += 1
        return self.family.apply_member(self.member, x, out, n_loc)
