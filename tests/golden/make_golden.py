#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own classes.

Run in the build container only (it needs /root/reference, which never
travels to the GPU box):

    python tests/golden/make_golden.py

Nothing of the reference is copied: this script imports its modules from
/root/reference, feeds them build-generated matrices and seeded vectors, and
stores inputs + outputs as data.

Two third-party modules the reference imports cannot be installed here, so
single-rank stand-ins (our own code, test helpers only) are registered before
the import:
* ``mpi4py``: COMM_WORLD with rank 0 / size 1;
* ``petsc4py``: ``Mat.SOR`` is served by the reference's OWN pure-Python
  ``Smoother`` (reference source/multigrid.py:83-97), so every multigrid
  golden vector is produced by reference arithmetic; what stays unpinned is
  PETSc's MatSOR itself.
"""
import importlib.util
import os
import sys
import time
import types
import warnings

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('STK_REFERENCE', '/root/reference')
PKG = os.path.join(REPO, 'spacetime-fullgrid-parallel_amd', 'source')

warnings.filterwarnings('ignore')


# ----------------------------------------------------------------------------
# stand-ins
# ----------------------------------------------------------------------------
class _Req:
    pass


class _Comm:
    def __init__(self, rank=0, size=1):
        self._rank, self._size = rank, size
        self.rank, self.size = rank, size
        self._queue = []

    def Get_rank(self):
        return self._rank

    def Get_size(self):
        return self._size

    def allreduce(self, x):
        return x

    def bcast(self, x, root=0):
        return x

    def gather(self, x, root=0):
        return [x]

    def Barrier(self):
        pass

    def Split_type(self, *a, **k):
        return self

    def Scatterv(self, send, recv):
        recv[...] = np.asarray(send[0]).reshape(recv.shape)

    def Gatherv(self, send, recv):
        recv[0][...] = np.asarray(send).reshape(recv[0].shape)

    def Isend(self, buf, dest=0, tag=0):
        self._queue.append(np.array(buf, copy=True))
        return _Req()

    def Recv(self, buf, source=0, tag=0):
        buf[...] = self._queue.pop(0).reshape(buf.shape)

    def Irecv(self, buf, source=0, tag=0):
        self.Recv(buf, source, tag)
        return _Req()


def install_standins():
    mpi4py = types.ModuleType('mpi4py')
    MPI = types.ModuleType('mpi4py.MPI')
    MPI.COMM_WORLD = _Comm()
    MPI.COMM_TYPE_SHARED = 0
    MPI.DOUBLE = 'double'
    MPI.Wtime = time.perf_counter

    class Request:
        @staticmethod
        def Waitall(reqs):
            pass

    MPI.Request = Request
    mpi4py.MPI = MPI
    sys.modules['mpi4py'] = mpi4py
    sys.modules['mpi4py.MPI'] = MPI

    petsc4py = types.ModuleType('petsc4py')
    PETSc = types.ModuleType('petsc4py.PETSc')
    PETSc.COMM_SELF = None

    class Vec:
        def createWithArray(self, arr, comm=None):
            self.arr = arr
            return self

        def setArray(self, a):
            self.arr[...] = a

    class Mat:
        class SORType:
            FORWARD_SWEEP = 'fwd'
            BACKWARD_SWEEP = 'bwd'

        def createAIJWithArrays(self, size, csr, comm=None):
            from source.multigrid import Smoother  # the reference's own
            indptr, indices, data = csr
            self.smoother = Smoother(
                sp.csr_matrix((data, indices, indptr), shape=size))
            return self

        def SOR(self, f, u, its, sortype):
            for _ in range(its):
                if sortype == Mat.SORType.FORWARD_SWEEP:
                    self.smoother.PreSmooth(u.arr, f.arr)
                else:
                    self.smoother.PostSmooth(u.arr, f.arr)

    PETSc.Vec, PETSc.Mat = Vec, Mat
    petsc4py.PETSc = PETSc
    sys.modules['petsc4py'] = petsc4py
    sys.modules['petsc4py.PETSc'] = PETSc


def load_build_module(name):
    """Our own mesh/assembly modules, loaded by path so that the name
    ``source`` stays free for the reference's package."""
    spec = importlib.util.spec_from_file_location('amd_' + name,
                                                  os.path.join(PKG, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def csr_parts(prefix, mat):
    mat = sp.csr_matrix(mat)
    return {
        prefix + '_indptr': mat.indptr.astype(np.int32),
        prefix + '_indices': mat.indices.astype(np.int32),
        prefix + '_data': mat.data.astype(np.float64),
        prefix + '_shape': np.array(mat.shape),
    }


class Hierarchy:
    """Duck-typed MeshHierarchy (reference multigrid.py:62-67)."""
    def __init__(self, P_mats):
        self.shared_comm = None
        self.J = len(P_mats)
        self.P_mats = P_mats
        self.R_mats = [P.T.tocsr() for P in P_mats]


def main():
    install_standins()
    sys.path.insert(0, REF)
    from source.lanczos import Lanczos
    from source.linalg import PCG
    from source.linop import CompositeLinOp, InvLinOp, KronLinOp
    from source.mpi_kron import (BlockDiagMPI, CompositeMPI,
                                 IdentityKronMatMPI, IdentityMPI,
                                 SparseKronIdentityMPI, SumMPI,
                                 TridiagKronIdentityMPI, TridiagKronMatMPI,
                                 as_matrix)
    from source.mpi_vector import DofDistributionMPI, KronVectorMPI
    from source.multigrid import MultiGrid
    from source.wavelets import (TransposedWaveletTransformKronIdentityMPI,
                                 WaveletTransformKronIdentityMPI,
                                 WaveletTransformOp)
    from mpi4py import MPI

    mesh_mod = load_build_module('mesh')
    mesh_mod.REFINE_NUMPY = True  # the generator does not load libstk
    asm = load_build_module('assembly')

    out_dir = os.environ.get('STK_GOLDEN_OUT', HERE)
    only = [k for k in os.environ.get('STK_GOLDEN_ONLY', '').split(',') if k]

    def save(name, **arrays):
        if only and name not in only:
            return
        path = os.path.join(out_dir, name + '.npz')
        np.savez_compressed(path, **arrays)
        print('wrote', os.path.relpath(path, REPO),
              '%.1f kB' % (os.path.getsize(path) / 1024))

    # ---- G1: wavelet transform matrices -----------------------------------
    out = {}
    for J in range(1, 6):
        for inter in (True, False):
            tag = 'J%d_%s' % (J, 'il' if inter else 'lv')
            op = WaveletTransformOp(J, interleaved=inter)
            out['W_' + tag] = as_matrix(op)
            out['WT_' + tag] = as_matrix(op.T)
            out['levels_' + tag] = np.asarray(op.levels)
            if inter:
                for j in range(0, J + 1):
                    out['split_%s_j%d' % (tag, j)] = op.split(j).toarray()
    save('g1_wavelets', **out)

    # ---- G2: time-slab partition tables -----------------------------------
    out = {}
    for N in (5, 9, 33, 65, 129):
        for size in (1, 2, 3, 4, 8):
            if size > N:
                continue
            for rank in (0, size - 1):
                d = DofDistributionMPI(_Comm(rank, size), N, 7)
                tag = 'N%d_s%d' % (N, size)
                out['dist_' + tag] = np.array(d.dof_distribution)
                out['counts_' + tag] = d.counts
                out['displs_' + tag] = d.displs
                out['dof2proc_' + tag] = d.dof2proc
                out['range_%s_r%d' % (tag, rank)] = np.array(
                    [d.t_begin, d.t_end])
    save('g2_partition', **out)

    # ---- problem data ------------------------------------------------------
    comm = MPI.COMM_WORLD

    def vec_from(dd, X):
        v = KronVectorMPI(dd)
        v.X_loc[:] = X
        return v

    for pname, meshfn, J_space, J_time in [
        ('square', mesh_mod.construct_2d_square_mesh, 2, 3),
        ('lshape', mesh_mod.construct_2d_lshape_mesh, 1, 3),
        ('square3', mesh_mod.construct_2d_square_mesh, 3, 2),
        ('cube', mesh_mod.construct_3d_cube_mesh, 1, 2),
    ]:
        if only and 'g3_' + pname not in only:
            continue
        mesh, _ = meshfn(J_space)
        tmesh = mesh_mod.construct_interval(2**J_time)
        A_t, L_t, M_t, G_t, u0_t = asm.time_matrices(tmesh)
        M_x, A_x = asm.space_matrices(mesh, scipy_path=True)  # the generator does not load libstk
        P_mats = asm.prolongation_matrices(mesh)
        u0_x = asm.space_load(
            mesh, lambda *c: np.prod([np.sin(np.pi * ck) for ck in c], axis=0), numpy_path=True)
        N, M = A_t.shape[0], M_x.shape[0]
        dd = DofDistributionMPI(comm, N, M)
        rng = np.random.RandomState(128)
        X = rng.rand(N, M)
        out = {'N': N, 'M': M, 'J_time': J_time, 'J_space': J_space, 'X': X,
               'u0_t': u0_t, 'u0_x': u0_x, 'nP': len(P_mats)}
        for nm, m in [('A_t', A_t), ('L_t', L_t), ('M_t', M_t), ('G_t', G_t),
                      ('M_x', M_x), ('A_x', A_x)]:
            out.update(csr_parts(nm, m))
        for j, P in enumerate(P_mats):
            out.update(csr_parts('P%d' % j, P))

        x = vec_from(dd, X)
        # G3: Kronecker applies
        for nm, T, S in [('AtMx', A_t, M_x), ('MtAx', M_t, A_x),
                         ('LtAx', L_t, A_x), ('LtTMx', L_t.T.tocsr(), M_x),
                         ('GtMx', G_t, M_x)]:
            x._invalidate()
            out['kron_' + nm] = (TridiagKronMatMPI(dd, T, S) @ x).X_loc
        x._invalidate()
        out['kron_metric'] = (SumMPI(dd, [
            TridiagKronMatMPI(dd, A_t, M_x),
            TridiagKronMatMPI(dd, M_t, A_x)
        ]) @ x).X_loc
        x._invalidate()
        out['tridiag_At'] = (TridiagKronIdentityMPI(dd, A_t) @ x).X_loc
        out['ident_Mx'] = (IdentityKronMatMPI(dd, M_x) @ x).X_loc
        out['sparse_At'] = (SparseKronIdentityMPI(dd, A_t) @ x).X_loc
        out['sparse_At_plusI'] = (SparseKronIdentityMPI(
            dd, A_t, add_identity=True) @ x).X_loc
        out['kronlinop_AtMx'] = KronLinOp(A_t, M_x) @ X.reshape(-1)

        # G4: wavelet transform in time (MPI composite form)
        W = WaveletTransformKronIdentityMPI(dd, J_time)
        WT = TransposedWaveletTransformKronIdentityMPI(dd, J_time)
        out['levels'] = np.asarray(W.levels)
        out['W'] = (W @ x).X_loc
        out['WT'] = (WT @ x).X_loc

        # G8: multigrid applies (reference MGM + reference Smoother)
        hier = Hierarchy(P_mats)
        b = rng.rand(M)
        out['mg_b'] = b
        for ss in (1, 3):
            for vc in (1, 2):
                mg = MultiGrid(A_x, hier, smoothsteps=ss, vcycles=vc)
                out['mg_Ax_s%d_v%d' % (ss, vc)] = mg @ b
        alpha = 0.3
        Cinv = [sp.csr_matrix(2**j * M_x + alpha * A_x)
                for j in range(J_time + 1)]
        mgC = MultiGrid(Cinv[2], hier, smoothsteps=3, vcycles=2)
        out['mg_C2_s3_v2'] = mgC @ b
        for j, A in enumerate(mg.mats):
            out.update(csr_parts('galerkin_Ax_%d' % j, A))

        # the full operator wiring of heateq_mpi.py:141-191
        for precond in ('direct', 'multigrid'):
            if precond == 'multigrid':
                mk = lambda m: MultiGrid(m, hier, smoothsteps=3, vcycles=2)
            else:
                mk = InvLinOp
            Kinv = mk(A_x)
            C_j = [mk(m) for m in Cinv]
            CAC_j = [CompositeLinOp([C_j[j], A_x, C_j[j]])
                     for j in range(J_time + 1)]
            S = SumMPI(dd, [
                TridiagKronMatMPI(dd, A_t, CompositeLinOp([M_x, Kinv, M_x])),
                TridiagKronMatMPI(dd, L_t, CompositeLinOp([M_x, Kinv, A_x])),
                TridiagKronMatMPI(dd, L_t.T.tocsr(),
                                  CompositeLinOp([A_x, Kinv, M_x])),
                TridiagKronMatMPI(dd, M_t, CompositeLinOp([A_x, Kinv, A_x])),
                TridiagKronMatMPI(dd, G_t, M_x),
            ])
            P = BlockDiagMPI(dd, [CAC_j[j] for j in W.levels])
            WT_S_W = CompositeMPI(dd, [WT, S, W])
            rhs = KronVectorMPI(dd)
            rhs.X_loc[:] = np.kron(u0_t, u0_x).reshape(-1, M)
            x._invalidate()
            out['S_' + precond] = (S @ x).X_loc
            out['P_' + precond] = (P @ x).X_loc  # G5
            out['WTSW_' + precond] = (WT_S_W @ x).X_loc

            # G6: PCG trajectory
            hist = []

            def cb(w, r, k):
                hist.append(r.dot(r))

            w, iters = PCG(WT_S_W, P, rhs, callback=cb)
            out['pcg_w_' + precond] = w.X_loc
            out['pcg_iters_' + precond] = iters
            out['pcg_rr_' + precond] = np.array(hist)
            w2, it2 = PCG(S, IdentityMPI(dd), rhs, kmax=60)
            out['pcg_unprec_w_' + precond] = w2.X_loc
            out['pcg_unprec_iters_' + precond] = it2

            # G7: Lanczos with a fixed start vector
            w0 = vec_from(dd, X)
            lz = Lanczos(WT_S_W, P, w=w0)
            out['lz_alpha_' + precond] = lz.alpha
            out['lz_beta_' + precond] = lz.beta
            out['lz_lmax_' + precond] = lz.lmax
            out['lz_lmin_' + precond] = lz.lmin
            out['lz_its_' + precond] = lz.iterations
        out['rhs'] = rhs.X_loc
        save('g3_' + pname, **out)


if __name__ == '__main__':
    main()
