"""Parallel-in-time heat-equation solver on MI355X GPUs (counterpart of the
reference's heateq_mpi.py; same class name, constructor arguments, operator
attributes and command line).

Launch one process per GPU:
    python heateq_mpi.py --J_time=5 --J_space=8
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        heateq_mpi.py --J_time=6 --J_space=9
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if __name__ == '__main__':
    # run as the driver, the allocator of the process is ours to set, and the time to do
    # it is before anything starts a thread (source/host_malloc.py)
    from source.host_malloc import keep_to_the_heap
    keep_to_the_heap()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from source import _lib, driver  # noqa: E402
from source.assembly import (prolongation_matrices, space_load,
                             space_matrices, time_matrices)
from source.comm import MPI
from source.linalg import PCG
from source.linop import (CompositeLinOp, EllMatrices, InvLinOp, as_space_op,
                          time_factor_steps)
from source.mpi_kron import (BlockDiagMPI, CompositeMPI, LinearOperatorMPI,
                             MatKronIdentityMPI, SumMPI, TridiagKronMatMPI,
                             _local_tridiag)
from source.mpi_vector import DofDistributionMPI, KronVectorMPI
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily
from source.problem import problem_helper
from source.wavelets import (TransposedWaveletTransformKronIdentityMPI,
                             WaveletTransformKronIdentityMPI,
                             WaveletTransformOp)


mem = driver.device_mb


class SchurMPI(LinearOperatorMPI):
    """The Schur complement S = B^T K B + G of the reference (sum of the five
    Kronecker terms of heateq_mpi.py:166-181), regrouped so that the spatial
    preconditioner K is applied twice instead of four times:

        u1 = (A_t kron M_x + L_t   kron A_x) x      one fused kernel
        u2 = (L_t^T kron M_x + M_t kron A_x) x      one fused kernel
        S x = (I kron M_x) K u1 + (I kron A_x) K u2 + (G_t kron M_x) x
                                                   one fused kernel
    This equals the five-term sum in exact arithmetic because K (a fixed number
    of V-cycles from zero) is linear; tests bound the difference.  The two K
    applies are independent: they run side by side on two HIP streams
    (MultiGrid.apply_pair; two_streams = False: one after the other)."""
    two_streams = True
    pack_last_stage = True  # False: the last stage in the plain sliced-ELL form (kron_ell.hip)

    def __init__(self, dofs_distr, A_t, L_t, M_t, G_t, M_x, A_x, Kinv_x):
        super().__init__(dofs_distr)
        self._factors = (A_t, L_t, M_t, G_t, M_x, A_x)
        self._linops = None
        self.Kinv_x = as_space_op(Kinv_x)
        self.ell = EllMatrices.shared([M_x, A_x])  # matrix 0 = M_x, 1 = A_x
        self._couples, self._steps = {}, {}

        def tri(T):
            host = _local_tridiag(dofs_distr, T)
            dev = _lib.to_dev(host)
            # does the factor reach the neighbour ranks' time rows at all?
            self._couples[id(dev)] = (host[0, 0] != 0.0, host[2, -1] != 0.0)
            self._steps[id(dev)] = time_factor_steps(host)
            return dev

        self.tA, self.tL, self.tM, self.tG = tri(A_t), tri(L_t), tri(
            M_t), tri(G_t)
        self.tLT = tri(L_t.T.tocsr())

    @property
    def linops(self):
        """The five Kronecker terms of the reference's SumMPI
        (heateq_mpi.py:166-181), built on first use: for callers that inspect
        ``S.linops`` (heateq_mpi_test.py:59-61, 120); the apply does not use
        them."""
        if self._linops is None:
            A_t, L_t, M_t, G_t, M_x, A_x = self._factors
            dd, K = self.dofs_distr, self.Kinv_x
            self._linops = [
                TridiagKronMatMPI(dd, A_t, CompositeLinOp([M_x, K, M_x])),
                TridiagKronMatMPI(dd, L_t, CompositeLinOp([M_x, K, A_x])),
                TridiagKronMatMPI(dd, L_t.T.tocsr(),
                                  CompositeLinOp([A_x, K, M_x])),
                TridiagKronMatMPI(dd, M_t, CompositeLinOp([A_x, K, A_x])),
                TridiagKronMatMPI(dd, G_t, M_x),
            ]
        return self._linops

    def _spec(self, tri, k, vec):
        """(tri, matrix, x, x_lo, x_hi) with the ghost rows only where the time
        factor couples to them."""
        lo, hi = self._couples[id(tri)]
        return (tri, k, vec.buf, vec.X_lo if lo else None,
                vec.X_hi if hi else None)

    def _matvec(self, vec_in, vec_out):
        assert (vec_in is not vec_out)
        self.time_communication = 0
        x = vec_in.buf
        n_loc, ld = vec_in.n_loc, vec_in.ld
        u = torch.empty_like(x)
        packed = self.ell.packed_for(n_loc)
        pair = getattr(self.Kinv_x, 'apply_pair', None) if self.two_streams else None
        if packed.ok:
            # packed matrix stream, ghost time steps fused into the one pass
            # (csrc/kron_pack.hip); the halo has to be there first
            ghosts = None
            first = [(self.tA, 0), (self.tL, 1)]
            from source.mpi_kron import _FusedKronSum
            if (self.dofs_distr.size > 1 and not vec_in.communicated_bdr
                    and _FusedKronSum.overlap and n_loc >= _FusedKronSum.OVERLAP_FROM):
                # the pass over the slab without the ghost steps while the halo is
                # in flight (reference mpi_kron.py:193-200), the two boundary steps
                # afterwards, from the compact records the pack leaves (_FusedKronSum.OVERLAP_FROM)
                self.time_communication = vec_in.communicate_bdr(
                    callback=lambda: packed.apply(first, x, None, n_loc, ld, 0.0, u), records=True)
                ghosts = vec_in.ghost_interleaved()
                packed.apply_boundary(first, vec_in.boundary_records(), ghosts, vec_in.X_lo is not None,
                                      vec_in.X_hi is not None, n_loc, ld, u)
            else:
                if self.dofs_distr.size > 1:
                    self.time_communication = vec_in.communicate_bdr()
                    ghosts = vec_in.ghost_interleaved()
                packed.apply(first, x, ghosts, n_loc, ld, 0.0, u)
            if pair is not None:
                # the second right-hand side and its K apply on the side stream,
                # beside the first K apply
                def second():
                    u2 = torch.empty_like(x)
                    packed.apply([(self.tLT, 0), (self.tM, 1)], x, ghosts, n_loc,
                                 ld, 0.0, u2)
                    return u2

                v1, v2 = pair(u, second, n_loc=n_loc, shared=(x, ghosts))
            else:
                v1 = self.Kinv_x.apply(u, n_loc=n_loc)
                packed.apply([(self.tLT, 0), (self.tM, 1)], x, ghosts, n_loc, ld,
                             0.0, u)
                v2 = self.Kinv_x.apply(u, n_loc=n_loc)
        else:
            first = [(self.tA, 0, x, None, None), (self.tL, 1, x, None, None)]
            if self.dofs_distr.size > 1:
                # slab-local part of the first launch while the halo is in flight
                self.time_communication = vec_in.communicate_bdr(
                    callback=lambda: self.ell.apply_local(first, n_loc, ld, 0.0, u))
            else:
                self.ell.apply_local(first, n_loc, ld, 0.0, u)
            self.ell.apply_ghost([self._spec(self.tA, 0, vec_in),
                                  self._spec(self.tL, 1, vec_in)], n_loc, ld, u)
            v1 = self.Kinv_x.apply(u, n_loc=n_loc)
            self.ell.apply([self._spec(self.tLT, 0, vec_in),
                            self._spec(self.tM, 1, vec_in)], n_loc, ld, 0.0, u)
            v2 = self.Kinv_x.apply(u, n_loc=n_loc)
        g_lo, g_hi = self._couples[id(self.tG)]
        if (self.pack_last_stage and packed.ok and not packed.explicit
                and not ((g_lo or g_hi) and self.dofs_distr.size > 1)):
            # the three inputs through the packed slot stream, one turn per term;
            # G_t's turn only in the lanes of the time steps it multiplies
            packed.apply_multi([(None, 0, v1), (None, 1, v2), (self.tG, 0, x)],
                               n_loc, ld, 0.0, vec_out.buf,
                               steps=[None, None, self._steps[id(self.tG)]])
        else:
            self.ell.apply([(None, 0, v1, None, None), (None, 1, v2, None, None),
                            self._spec(self.tG, 0, vec_in)], n_loc, ld, 0.0,
                           vec_out.buf)
        vec_out.communicated_bdr = False
        return vec_out


class _Beside:
    """fn(*args) in a thread of its own (host work that holds no interpreter lock
    for long: NumPy on large arrays); result() joins and re-raises.  A daemon
    thread: nothing to shut down if the caller never gets to result()."""
    def __init__(self, fn, *args):
        import threading
        self._out = self._err = None

        def run():
            try:
                self._out = fn(*args)
            except BaseException as err:  # handed to the caller of result()
                self._err = err

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def result(self):
        self._thread.join()
        if self._err is not None:
            raise self._err
        return self._out


class HeatEquationMPI:
    """Creates the operators for solving the heat equation in parallel in time
    (reference heateq_mpi.py:29-201).  Matrices come from the build's own P1
    assembly (source/assembly.py) instead of NGSolve; every rank assembles them
    itself and uploads its own copy to its GPU (the reference shares them
    through MPI-3 windows, which has no meaning across HBMs).

    schur='fused' (default) builds S as SchurMPI; schur='reference' builds the
    five-term SumMPI exactly as reference heateq_mpi.py:166-181.

    family='batched' (default) applies the preconditioner's multigrids for
    2^j M_x + alpha A_x, all j, as one batched V-cycle over a shared pair of
    hierarchies; family='reference' builds one MultiGrid per j from the assembled
    matrix, as reference heateq_mpi.py:147-153 does.

    arithmetic='accurate' (default) keeps the fast structure (two multigrid applies
    per S on two streams, one batched V-cycle for P) with the reference's
    arithmetic in the two places that own the gap to the CPU path's r.Pr history
    (DESIGN.md section 5), where they own it -- the last V-cycle of a multigrid
    application on the finest level: the restricted residual as R (A u - f)
    (multigrid.py:174-175) and the post-smoothing on Gauss-Seidel rows with their
    diagonal, u_i += (f_i - row_i u) / a_ii (multigrid.py:89-97).  Every entry of
    the history within 1e-10 of the CPU path -- the bound BASELINE.json's north
    star states -- (measured <= 4.3e-11 at configs 1-4) for 4 % more solve time
    than the fast mode.

    arithmetic='fast': diagonal-free Gauss-Seidel rows u_i = (f_i - sum_{j != i}) / a_ii
    and the restricted residual as (R A) u - R f from the precomputed product on every
    level and in every V-cycle: 4 % less solve time, iteration counts unchanged, history entries within 4.6e-10.

    arithmetic='reference' = schur='reference' + family='reference' + Gauss-Seidel
    rows with their diagonal (u_i += (f_i - row_i u) / a_ii, multigrid.py:89-97) +
    the restricted residual as R (A u - f) (multigrid.py:174-175) instead of
    (R A) u - R f: every regrouping this build adds is switched off, what remains
    against the CPU path is the order of additions inside dot products and fused
    multiply-adds."""
    def __init__(self,
                 J_space=2,
                 J_time=None,
                 problem='square',
                 wavelettransform='composite',
                 precond='multigrid',
                 smoothsteps=3,
                 alpha=0.3,
                 vcycles=2,
                 schur='fused',
                 family='batched',
                 arithmetic='accurate',
                 comm=None):
        # the set-up's large host temporaries come from the heap (source/_lib.py: the
        # planner threads otherwise queue on the address-space lock of the process)
        with _lib.host_heap_for_setup():
            self._set_up(J_space, J_time, problem, wavelettransform, precond, smoothsteps, alpha,
                         vcycles, schur, family, arithmetic, comm)

    def _set_up(self, J_space, J_time, problem, wavelettransform, precond, smoothsteps, alpha,
                vcycles, schur, family, arithmetic, comm):
        start_time = MPI.Wtime()
        # (label, seconds since the start) of the stages of the set-up, for
        # tools/setup_profile.py --timeline
        self.setup_timeline = []
        mark = lambda label: self.setup_timeline.append((label, MPI.Wtime() - start_time))
        # (label, begin, end) of what runs in side threads
        self.setup_threads = []

        def timed(label, fn):
            def run(*args, **kw):
                begin = MPI.Wtime() - start_time
                try:
                    return fn(*args, **kw)
                finally:
                    self.setup_threads.append((label, begin, MPI.Wtime() - start_time))
            return run
        comm = MPI.COMM_WORLD if comm is None else comm
        assert arithmetic in ('fast', 'accurate', 'reference')
        assert family in ('batched', 'reference'), family
        assert schur in ('fused', 'reference'), schur
        if arithmetic == 'reference':
            schur = family = 'reference'
        self.arithmetic = arithmetic
        if J_time is None:
            J_time = J_space
        self.J_time = J_time
        self.J_space = J_space
        self.alpha = alpha

        mesh_space, bc_space, mesh_time, data, fn = problem_helper(
            problem, J_space=J_space, J_time=J_time)
        mark('meshes')
        # the load vector and the prolongations need the mesh only: beside the
        # assembly, which runs on the host threads of libstk (no GIL held)
        from concurrent.futures import ThreadPoolExecutor
        u0_x = _Beside(timed('load vector', space_load), mesh_space, data['u0'])
        # ... and the hierarchy: the object (prolongations) is published as soon as it
        # exists; the same thread then works ahead on what the plans share (tile orders,
        # bands, transfer operators on the device) -- plans ask through hierarchy.shared(),
        # made by whoever asks first
        import threading
        published, box = threading.Event(), {}

        def build_hierarchy(mesh):
            try:
                box['hierarchy'] = timed('prolongations', MeshHierarchy)(mesh)
            finally:
                published.set()
            if precond == 'multigrid':
                timed('shares of the hierarchy', _lib.in_device_context(box['hierarchy'].prepare))()

        shares = _Beside(build_hierarchy, mesh_space)
        # --- TIME --- (heateq_mpi.py:78-88)
        self.A_t, self.L_t, self.M_t, self.G_t, self.u0_t = time_matrices(
            mesh_time)
        # --- SPACE --- (heateq_mpi.py:91-98)
        self.M_x, self.A_x = space_matrices(mesh_space)
        mark('time and space matrices')
        self.N = self.A_t.shape[0]
        self.M = self.M_x.shape[0]
        assert (len(data['g']) == 0)
        self.dofs_distr = DofDistributionMPI(comm, self.N, self.M)

        # --- Wavelet transform --- (heateq_mpi.py:126-139)
        if wavelettransform == 'composite':
            self.W = WaveletTransformKronIdentityMPI(self.dofs_distr,
                                                     self.J_time)
            self.WT = TransposedWaveletTransformKronIdentityMPI(
                self.dofs_distr, self.J_time)
        elif wavelettransform == 'original':
            self.W_t = WaveletTransformOp(self.J_time)
            self.W = MatKronIdentityMPI(self.dofs_distr, self.W_t)
            self.WT = MatKronIdentityMPI(self.dofs_distr, self.W_t.T)
        elif wavelettransform == 'interleaved':
            self.W_t = WaveletTransformOp(self.J_time, interleaved=True)
            self.W = MatKronIdentityMPI(self.dofs_distr, self.W_t)
            self.WT = MatKronIdentityMPI(self.dofs_distr, self.W_t.T)
        else:
            raise ValueError(wavelettransform)

        # ---- Preconditioners in space ---- (heateq_mpi.py:141-162)
        published.wait()
        if 'hierarchy' not in box:
            shares.result()  # raises what the constructor raised
        hierarchy = box['hierarchy']
        mark('wavelets, prolongations')
        self.hierarchy = hierarchy
        # Row form of the Gauss-Seidel copies.  'owned': the reference's forms where they
        # matter -- the gap of the fast mode is owned by the FINEST level
        # (profiles/r03_history_by_level.log) and there by the LAST V-cycle's restricted
        # residual and post-smoothing (profiles/r03_history_by_cycle.log); what comes
        # earlier is damped by what follows it.  The finest level gets both row forms.
        gs_rows = {'reference': 'full', 'accurate': self.ACCURATE['gs_rows'], 'fast': None}[arithmetic]
        if precond == 'multigrid' and family == 'reference':
            # one hierarchy per wavelet level from the assembled matrix
            # (reference heateq_mpi.py:147-153)
            fuse = False if arithmetic == 'reference' else None
            self.Kinv_x = MultiGrid(self.A_x, hierarchy, smoothsteps=smoothsteps,
                                    vcycles=vcycles, fuse_restrict=fuse, gs_rows=gs_rows)
            self.C_family = None
            self.C_j = [
                MultiGrid(2**j * self.M_x + alpha * self.A_x, hierarchy,
                          smoothsteps=smoothsteps, vcycles=vcycles,
                          fuse_restrict=fuse, gs_rows=gs_rows)
                for j in range(self.J_time + 1)
            ]
            if arithmetic == 'accurate':
                for mg in [self.Kinv_x] + self.C_j:
                    self._accurate_options(mg._dev, hierarchy.J, vcycles)
        elif precond == 'multigrid':
            # the two hierarchies (A_x alone; 2^j M_x + alpha A_x, all j in one
            # family) are independent host work (SciPy / NumPy release the GIL)
            # worker threads start on device 0: pin them to this rank's GPU
            on_dev = _lib.in_device_context
            with ThreadPoolExecutor(max_workers=4) as pool:
                if schur != 'reference':  # the Kronecker plan S streams: independent of both
                    n_steps = self.dofs_distr.t_end - self.dofs_distr.t_begin
                    pool.submit(timed('Kronecker plan', on_dev(lambda: EllMatrices.shared(
                        [self.M_x, self.A_x]).packed_for(n_steps))))
                kinv = pool.submit(timed('plan of K', on_dev(MultiGrid)), self.A_x, hierarchy,
                                   smoothsteps=smoothsteps, vcycles=vcycles, gs_rows=gs_rows)
                # bands of 6 mesh rows for the family's strip-wise sweeps on longer slabs
                # (source/multigrid.py BAND_MERGE: P -2 %, bit-identical); an
                # environment override serves the A/B
                n_steps_ = self.dofs_distr.t_end - self.dofs_distr.t_begin
                merge = None if 'STK_BAND_MERGE_FAMILY' in os.environ else (6 if n_steps_ >= 16 else 1)
                members = pool.submit(
                    timed('plan of the family', on_dev(MultiGridFamily)), self.A_x, self.M_x, hierarchy, ca=alpha,
                    cms=[2**j for j in range(self.J_time + 1)],
                    smoothsteps=smoothsteps, vcycles=vcycles, gs_rows=gs_rows, band_merge=merge,
                    exact_coarse=(arithmetic == 'accurate' and self.ACCURATE['member_coarse_matrices']))
                self.Kinv_x, self.C_family = kinv.result(), members.result()
            if arithmetic == 'accurate':
                for plans in (self.Kinv_x._dev, self.C_family._dev):
                    self._accurate_options(plans, hierarchy.J, vcycles)
            # strips of the strip-wise smoothing (csrc/mg.hip), measured at config 3
            # (profiles/r03_strip_sizes_two_streams.log): K's applies run two at a time
            # inside S and share the caches, the family's applies run alone
            self.Kinv_x._dev.set_option('strip_pct', 60)
            self.C_family._dev.set_option('strip_pct', 160)
            self.C_j = self.C_family.members
        else:
            assert (precond == 'direct')
            self.Kinv_x = InvLinOp(self.A_x)
            self.C_j = [
                InvLinOp(2**j * self.M_x + alpha * self.A_x)
                for j in range(self.J_time + 1)
            ]
        self.u0_x = u0_x.result()
        shares.result()
        mark('multigrid plans, Kronecker plan, load vector')
        self.CAC_j = [
            CompositeLinOp([self.C_j[j], self.A_x, self.C_j[j]])
            for j in range(self.J_time + 1)
        ]

        # -- MPI objects -- (heateq_mpi.py:164-185)
        dd = self.dofs_distr
        if schur == 'reference':
            M_x, A_x, K = self.M_x, self.A_x, self.Kinv_x
            self.A_MKM = TridiagKronMatMPI(dd, self.A_t,
                                           CompositeLinOp([M_x, K, M_x]))
            self.L_MKA = TridiagKronMatMPI(dd, self.L_t,
                                           CompositeLinOp([M_x, K, A_x]))
            self.LT_AKM = TridiagKronMatMPI(dd, self.L_t.T.tocsr(),
                                            CompositeLinOp([A_x, K, M_x]))
            self.M_AKA = TridiagKronMatMPI(dd, self.M_t,
                                           CompositeLinOp([A_x, K, A_x]))
            self.G_M = TridiagKronMatMPI(dd, self.G_t, M_x)
            self.S = SumMPI(
                dd,
                [self.A_MKM, self.L_MKA, self.LT_AKM, self.M_AKA, self.G_M])
        else:
            self.S = SchurMPI(dd, self.A_t, self.L_t, self.M_t, self.G_t,
                              self.M_x, self.A_x, self.Kinv_x)

        levels = (self.W.levels if hasattr(self.W, 'levels') else
                  WaveletTransformOp(self.J_time, interleaved=True).levels)
        self.P = BlockDiagMPI(dd, [self.CAC_j[j] for j in levels])
        if precond == 'multigrid' and family != 'reference' and schur != 'reference':
            self.P.mid_packed = (self.S.ell, 1)  # A_x in S's packed (M_x, A_x) stream
        self.WT_S_W = CompositeMPI(dd, [self.WT, self.S, self.W])

        # -- RHS -- (heateq_mpi.py:188-191)
        self.rhs = KronVectorMPI(dd)
        # u0_t kron u0_x, formed on the device (stk_outer): one product per entry,
        # the same doubles as np.kron on the host
        u_t = _lib.to_dev(self.u0_t[self.rhs.t_begin:self.rhs.t_end])
        u_x = _lib.to_dev(self.u0_x)
        _lib.check(_lib.lib().stk_outer(
            _lib.stream(), self.M, self.rhs.n_loc, self.rhs.ld, _lib.ptr(u_t),
            _lib.ptr(u_x), _lib.ptr(self.rhs.buf)))

        from source.linop import forget_union_pattern
        forget_union_pattern()
        if hasattr(self.hierarchy, 'forget'):
            self.hierarchy.forget()  # Galerkin chains, R A products: only the plans' construction needs them
        mark('operators and right-hand side')
        self.setup_time = MPI.Wtime() - start_time
        self.mem_after_mpi = mem()

    # What arithmetic='accurate' switches back to the reference's forms (class-level so
    # that tools/history_attribution.py can move one knob at a time): the row form of
    # the Gauss-Seidel copies, up to which level the restricted residual stays fused,
    # how many leading V-cycles keep the fast forms, and which parts of the last one.
    ACCURATE = {'gs_rows': 'owned', 'fuse_restrict_below_finest': True, 'fast_leading_cycles': True,
                'fast_parts': 1, 'member_coarse_matrices': True}

    @classmethod
    def _accurate_options(cls, plans, J, vcycles):
        """The plan options of arithmetic='accurate' (gs_rows='owned' plans): the
        restricted residual as R (A u - f) (reference multigrid.py:174-175) on the
        finest level, (R A) u - R f below it; the fast forms in all but the last
        V-cycle and in the last one's pre-smoothing."""
        acc = cls.ACCURATE
        plans.set_option('fuse_restrict_max_level', J - 1 if acc['fuse_restrict_below_finest'] else -1)
        plans.set_option('fast_until_cycle', vcycles - 1 if acc['fast_leading_cycles'] else 0)
        plans.set_option('fast_parts', acc['fast_parts'])

    def print_time_per_apply(self):
        for name in driver.OPERATORS:
            print('%-4s%.5f\t%.5f' % ((name + ':',) + tuple(getattr(self, name).time_per_apply())))
        print('')


def main(argv=None):
    args = driver.parse('Solve heatequation on MI355X GPUs, one time slab each.', argv,
                        extra=[('schur', str, 'fused',
                                'fused (2 multigrid applies per S) or reference'),
                               ('arithmetic', str, 'accurate',
                                'accurate: Gauss-Seidel and restricted residual in the '
                                'reference\'s arithmetic (r.Pr history within 1e-10 of the CPU '
                                'path); fast: both regrouped everywhere (4 %% less solve time, history '
                                'within 4.6e-10); reference: every regrouping of the build off '
                                '(2.3x slower than fast)')])
    comm, rank, size = driver.start(args)
    heat = HeatEquationMPI(**driver.solver_arguments(args))
    # per-rank record, gathered and printed as one blob at the end
    record = {'rank': rank, 'mem_after_construction': mem()}
    if size > 1:
        # first contact with several GPUs: where this rank's plans live, the backend and
        # RCCL version, peer access (stderr); the halo form -- direct, or with
        # STK_HALO_ROUTES=auto chosen by a probe on real rows among direct and routed
        # over 3 / 7 links (source/mpi_vector.py choose_halo_form)
        from source.mpi_vector import choose_halo_form, startup_report
        record['startup'] = startup_report(heat.dofs_distr, [heat.rhs.buf])
        record['halo_form'] = choose_halo_form(heat.dofs_distr)
    if rank == 0:
        record.update(args=vars(args), N=heat.N, M=heat.M)
        driver.report_construction(heat)

    def progress(w, residual, k):
        if rank == 0:
            print('.', end='', flush=True)

    LinearOperatorMPI.sync_timing = True
    type(comm).timing = size > 1  # count and time the scalar all-reduces of the solve
    comm.reset_counters()
    comm.Barrier()
    began = MPI.Wtime()
    history = []
    solution, iters = PCG(heat.WT_S_W, heat.P, heat.rhs, callback=progress,
                          history=history)
    comm.Barrier()
    record.update(solve_time=MPI.Wtime() - began, mem_after_solve=mem(),
                  iters=iters, r_dot_Pr=list(history),
                  allreduce_calls=comm.allreduce_calls,
                  allreduce_host_s=comm.allreduce_host_s,
                  # as the reference's time_applies (mpi_kron.py:23-31) every apply is
                  # bracketed by a device synchronisation, and on several ranks every
                  # scalar all-reduce by two: solve_time is taken WITH them (bench.py
                  # times its iterations without)
                  solve_time_includes='per-apply and per-all-reduce timing synchronisations')
    type(comm).timing = False
    for name in driver.OPERATORS + ('WT_S_W',):
        record[name] = driver.counters(getattr(heat, name))
    if rank == 0:
        print('\nCompleted in %d PCG steps.' % iters)
        print('Total solve time: %ss.' % record['solve_time'])
        print('Final r.Pr: %s' % history[-1])
        heat.print_time_per_apply()
        print('Device memory after solve: %smb.' % mem())
    driver.publish(comm, record)
    return heat, solution, iters, history


if __name__ == "__main__":
    main()
