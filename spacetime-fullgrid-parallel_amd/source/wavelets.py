"""Wavelet-in-time transform (counterpart of reference source/wavelets.py).

W maps 3-point wavelet coordinates to hat-function coordinates in time.  On a
single rank the whole transform is one fused HIP kernel
(``stk_wavelet_apply``: all J levels in LDS, one read and one write of the
vector); across ranks it is the reference's composite of per-level
``SparseKronIdentityMPI(split(j), add_identity=True)`` factors
(reference wavelets.py:172-198), whose strided partner exchange runs over
torch.distributed.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from .mpi_kron import CompositeMPI, SparseKronIdentityMPI
from .mpi_vector import KronVectorMPI


def _level_matrix(J, j):
    """I + split(j): the level-j step on the stride-2^(J-j) nodes of the
    interleaved numbering, as a sparse (2^J+1)^2 matrix.
    odd k:  y = 1/2 (x[k-1] + x[k+1]) + s x[k];
    even k: y = x[k] - 1/2 s (x[k-1] + x[k+1]), -s at the two end nodes
    (reference wavelets.py:81-104, 136-169)."""
    n = 2**J + 1
    S, s, nk = 2**(J - j), 2**(j / 2), 2**j
    rows, cols, vals = [], [], []
    for k in range(nk + 1):
        r = k * S
        if k % 2:
            rows += [r, r, r]
            cols += [r - S, r, r + S]
            vals += [0.5, s, 0.5]
        else:
            rows.append(r), cols.append(r), vals.append(1.0)
            for nb in (k - 1, k + 1):
                if 0 <= nb <= nk:
                    end = k == 0 or k == nk
                    rows.append(r), cols.append(nb * S)
                    vals.append(-s if end else -0.5 * s)
    other = np.setdiff1d(np.arange(n), np.arange(0, n, S))
    rows += list(other)
    cols += list(other)
    vals += [1.0] * len(other)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, n))


def WaveletTransformMat(J):
    """The transform as an explicit matrix, wavelets ordered level by level
    (debug helper of the reference, wavelets.py:9-42; dense-ish, host only).
    Built from the two-scale relations directly, independently of the device
    kernel: T_0 = I_2 and T_j = [P_j T_{j-1} | Q_j], where P_j interpolates hat
    functions from 2^(j-1) to 2^j elements and column m of Q_j is the 3-point
    wavelet at the odd node 2m+1: 2^(j/2) * (-1/2, 1, -1/2), with -1 instead
    of -1/2 on a boundary node."""
    T = np.eye(2)
    for j in range(1, J + 1):
        nc, nf = 2**(j - 1) + 1, 2**j + 1
        P = np.zeros((nf, nc))
        P[np.arange(0, nf, 2), np.arange(nc)] = 1.0
        odd = np.arange(1, nf, 2)
        P[odd, odd // 2] = 0.5
        P[odd, odd // 2 + 1] = 0.5
        Q = np.zeros((nf, nc - 1))
        cols = np.arange(nc - 1)
        Q[odd, cols] = 1.0
        Q[odd - 1, cols] = -0.5
        Q[odd + 1, cols] = -0.5
        Q[0, 0] = Q[-1, -1] = -1.0
        T = np.hstack([P @ T, 2**(j / 2) * Q])
    return sp.csr_matrix(T)


class WaveletTransformOp(sp.linalg.LinearOperator):
    """Matrix-free W_t, applied on the device (reference wavelets.py:45-169).

    interleaved=False: wavelets ordered level by level [d_0, d_1, ..., d_J];
    interleaved=True: numbered by their node on the finest mesh."""
    def __init__(self, J, interleaved=False):
        super().__init__(dtype=np.float64, shape=(2**J + 1, 2**J + 1))
        self.J = J
        self.interleaved = interleaved
        n = 2**J + 1
        lv = np.zeros(n, dtype=int)
        for j in reversed(range(0, J + 1)):
            lv[::2**(J - j)] = j
        if interleaved:
            self.levels = lv
            self._pos = None
        else:
            # position in the interleaved numbering of the k-th level-ordered
            # wavelet: level 0 -> the two end nodes, level j -> odd multiples
            # of 2^(J-j)
            pos = [0, n - 1]
            for j in range(1, J + 1):
                S = 2**(J - j)
                pos += [(2 * m + 1) * S for m in range(2**(j - 1))]
            self._pos = np.array(pos)
            self.levels = [int(lv[p]) for p in pos]

    def _device_apply(self, X, transposed):
        X = np.asarray(X, dtype=np.float64)
        n = self.shape[0]
        X2 = X.reshape(n, -1)
        x = _lib.to_dev(np.ascontiguousarray(X2.T))  # (k, n): time contiguous
        y = torch.empty_like(x)
        _lib.check(_lib.lib().stk_wavelet_apply(_lib.stream(), x.shape[0],
                                                self.J, n, int(transposed),
                                                _lib.ptr(x), _lib.ptr(y)))
        return y.cpu().numpy().T.reshape(X.shape)

    def _matmat(self, X):
        X = np.asarray(X, dtype=np.float64)
        if self._pos is not None:
            Xi = np.empty_like(X)
            Xi[self._pos] = X
            X = Xi
        return self._device_apply(X, False)

    def _rmatmat(self, X):
        Y = self._device_apply(X, True)
        if self._pos is not None:
            Y = Y[self._pos]
        return Y

    def _matvec(self, x):
        return self._matmat(np.asarray(x).reshape(-1, 1)).reshape(-1)

    def _rmatvec(self, x):
        return self._rmatmat(np.asarray(x).reshape(-1, 1)).reshape(-1)

    def split(self, j):
        """The level-j step minus the identity, [p(j) q(j)] - I on the level-j
        nodes, such that W = prod_j (I + split(j)) (reference
        wavelets.py:136-169)."""
        n = 2**self.J + 1
        if j == 0:
            return sp.csr_matrix((n, n))
        S = 2**(self.J - j)
        full = _level_matrix(self.J, j) - sp.identity(n, format='csr')
        # keep the explicit -1 + 1 = 0 free structure of the reference: entries
        # only on the level-j nodes, including the (zero-sum) diagonal there
        full = sp.csr_matrix(full)
        keep = np.zeros(n, dtype=bool)
        keep[::S] = True
        coo = full.tocoo()
        m = keep[coo.row]
        out = sp.csr_matrix((coo.data[m], (coo.row[m], coo.col[m])),
                            shape=(n, n))
        return out


class _FusedWavelet:
    def __init__(self, J, transposed):
        self.J, self.transposed = J, transposed

    def apply(self, vec_in, vec_out):
        _lib.check(_lib.lib().stk_wavelet_apply(
            _lib.stream(), vec_in.M, self.J, vec_in.ld, int(self.transposed),
            _lib.ptr(vec_in.buf), _lib.ptr(vec_out.buf)))


class _TransposedWavelet:
    """W or W^T across ranks through two all-to-all transposes: every rank
    collects ALL time steps of its share of the space dofs (rows
    [x_begin, x_end) of every rank's slab), runs the fused single-rank kernel on
    them, and sends the result back.  On a fully connected xGMI node an
    all-to-all moves 1/size of the slab over each link at once, whereas the
    per-level composite needs J dependent exchanges of whole time rows
    (reference wavelets.py:172-198, mpi_vector.py:189-203)."""
    def __init__(self, dofs_distr, J, transposed):
        from .mpi_vector import DofDistributionMPI
        self.dd = dofs_distr
        self.J, self.transposed = J, transposed
        self.space = DofDistributionMPI(dofs_distr.comm, dofs_distr.M, 1)

    def apply(self, vec_in, vec_out):
        from .comm import MPI
        dd, comm = self.dd, self.dd.comm
        rank, N = dd.rank, dd.N
        xb, xe = self.space.dof_distribution[rank]
        ldN = N + (N & 1)
        full = torch.zeros((xe - xb, ldN), dtype=torch.float64,
                           device=vec_in.buf.device)
        t0 = MPI.Wtime()
        sends, recvs, parts = [], [], []
        n_in, ld_in = vec_in.n_loc, vec_in.ld

        def block(src, rows, cols, ld_src, dst, ld_dst, src_off=0, dst_off=0):
            # a block of rows moved to another leading dimension: libstk's
            # stk_copy_block on the device (host tensors: CPU tests of the layer)
            if src.is_cuda:
                _lib.copy_block(src, rows, cols, ld_src, dst, ld_dst, src_off, dst_off)
            elif rows and cols:
                dst.view(-1)[dst_off:].as_strided((rows, cols), (ld_dst, 1)).copy_(
                    src.view(-1)[src_off:].as_strided((rows, cols), (ld_src, 1)))

        for p in range(comm.size):
            pb, pe = self.space.dof_distribution[p]
            tb, te = dd.dof_distribution[p]
            if p == rank:
                block(vec_in.buf, xe - xb, n_in, ld_in, full, ldN, xb * ld_in, tb)
                continue
            sbuf = torch.empty((pe - pb, n_in), dtype=torch.float64, device=full.device)
            block(vec_in.buf, pe - pb, n_in, ld_in, sbuf, n_in, pb * ld_in)
            sends.append((sbuf, p))
            r = torch.empty((xe - xb, te - tb), dtype=torch.float64,
                            device=full.device)
            recvs.append((r, p))
            parts.append((tb, te, r))
        comm.wait_all(comm.exchange(sends, recvs))
        for tb, te, r in parts:
            block(r, xe - xb, te - tb, te - tb, full, ldN, 0, tb)
        t_comm = MPI.Wtime() - t0
        out = torch.empty_like(full)
        if xe > xb:
            _lib.check(_lib.lib().stk_wavelet_apply(
                _lib.stream(), xe - xb, self.J, ldN, int(self.transposed),
                _lib.ptr(full), _lib.ptr(out)))
        t0 = MPI.Wtime()
        sends, recvs, parts = [], [], []
        tb_me, te_me = dd.dof_distribution[rank]
        for p in range(comm.size):
            pb, pe = self.space.dof_distribution[p]
            tb, te = dd.dof_distribution[p]
            if p == rank:
                block(out, xe - xb, te - tb, ldN, vec_out.buf, vec_out.ld, tb,
                      xb * vec_out.ld)
                continue
            sbuf = torch.empty((xe - xb, te - tb), dtype=torch.float64, device=full.device)
            block(out, xe - xb, te - tb, ldN, sbuf, te - tb, tb)
            sends.append((sbuf, p))
            r = torch.empty((pe - pb, te_me - tb_me), dtype=torch.float64,
                            device=full.device)
            recvs.append((r, p))
            parts.append((pb, pe, r))
        comm.wait_all(comm.exchange(sends, recvs))
        n_out = vec_out.n_loc
        for pb, pe, r in parts:
            block(r, pe - pb, n_out, n_out, vec_out.buf, vec_out.ld, 0, pb * vec_out.ld)
        if vec_out.ld > vec_out.n_loc:
            vec_out.buf[:, vec_out.n_loc:].zero_()
        return t_comm + MPI.Wtime() - t0


class WaveletTransformKronIdentityMPI(CompositeMPI):
    """W := W_t kron Id_x (reference wavelets.py:172-183)."""
    def __init__(self, dofs_distr, J):
        wavelet_transform = WaveletTransformOp(J, interleaved=True)
        self.levels = wavelet_transform.levels
        linops = []
        for j in reversed(range(1, J + 1)):
            split_mat = wavelet_transform.split(j)
            linops.append(
                SparseKronIdentityMPI(dofs_distr, split_mat,
                                      add_identity=True))
        super().__init__(dofs_distr, linops)
        self._fused = _FusedWavelet(J, False) if dofs_distr.size == 1 else None
        # across ranks: 'transpose' (default) or 'composite' (the reference's
        # per-level exchange)
        # (with fewer space dofs than ranks there is nothing to transpose onto)
        self.mode = 'transpose' if dofs_distr.M >= dofs_distr.size else 'composite'
        self._transposed = (_TransposedWavelet(dofs_distr, J, False)
                            if self.mode == 'transpose' else None)

    def _matvec(self, vec_in, vec_out):
        if self._fused is None and self.mode == 'composite':
            return super()._matvec(vec_in, vec_out)
        if self._fused is None:
            self.time_communication = self._transposed.apply(vec_in, vec_out)
        else:
            self.time_communication = 0
            self._fused.apply(vec_in, vec_out)
        vec_out.communicated_bdr = False
        return vec_out


class TransposedWaveletTransformKronIdentityMPI(CompositeMPI):
    """W.T := W_t.T kron Id_x (reference wavelets.py:186-198)."""
    def __init__(self, dofs_distr, J):
        wavelet_transform = WaveletTransformOp(J, interleaved=True)
        self.levels = wavelet_transform.levels
        linops = []
        for j in range(1, J + 1):
            split_mat = wavelet_transform.split(j)
            linops.append(
                SparseKronIdentityMPI(dofs_distr,
                                      split_mat.T.tocsr(),
                                      add_identity=True))
        super().__init__(dofs_distr, linops)
        self._fused = _FusedWavelet(J, True) if dofs_distr.size == 1 else None
        # (with fewer space dofs than ranks there is nothing to transpose onto)
        self.mode = 'transpose' if dofs_distr.M >= dofs_distr.size else 'composite'
        self._transposed = (_TransposedWavelet(dofs_distr, J, True)
                            if self.mode == 'transpose' else None)

    def _matvec(self, vec_in, vec_out):
        if self._fused is None and self.mode == 'composite':
            return super()._matvec(vec_in, vec_out)
        if self._fused is None:
            self.time_communication = self._transposed.apply(vec_in, vec_out)
        else:
            self.time_communication = 0
            self._fused.apply(vec_in, vec_out)
        vec_out.communicated_bdr = False
        return vec_out
