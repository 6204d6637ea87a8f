"""Time-slab distributed vectors on the GPU (counterpart of reference
source/mpi_vector.py).

Same classes, attributes and call signatures as the reference; differences:
* the slab lives in HBM, space-major: ``_buf[i, t]`` (shape (M, ld)), so that a
  time column is contiguous (see include/stk.h).  ``X_loc`` is the reference's
  (N_loc, M) view of it (a transposed torch view, writable);
* arithmetic runs in libstk's BLAS-1 kernels; ``alpha * p`` is lazy so that the
  reference's ``w += alpha * p`` (linalg.py:29) becomes one fused pass;
* communication goes through torch.distributed (RCCL/xGMI) instead of mpi4py.
"""
import threading
import weakref

import numpy as np
import torch

from . import _lib
from .comm import MPI


class DofDistributionMPI:
    """Block partition of the N time dofs over the ranks: N // size each, the
    last N % size ranks get one more (reference mpi_vector.py:5-38)."""
    def __init__(self, comm, N, M):
        self.N = N
        self.M = M
        self.comm = comm
        self.rank = comm.Get_rank()
        self.size = comm.Get_size()
        assert (self.N >= self.size)

        block_size, rest_size = divmod(self.N, self.size)
        self.dof_distribution = []
        self.displs = np.empty(self.size)
        self.counts = np.empty(self.size)
        start = 0
        for p in range(self.size):
            stop = start + block_size + (1 if self.size - p - 1 < rest_size
                                         else 0)
            self.dof_distribution.append([start, stop])
            self.displs[p] = start * self.M
            self.counts[p] = (stop - start) * self.M
            start = stop
        assert (start == self.N)
        self.t_begin, self.t_end = self.dof_distribution[self.rank]

        self.dof2proc = np.zeros(self.N)
        for p, (t_begin, t_end) in enumerate(self.dof_distribution):
            self.dof2proc[t_begin:t_end] = p


def _fill_slab(buf, n_loc, src):
    """buf[i, t] = src[t, i] for a time-major (n_loc, M) block `src` (padding
    columns zero).  On the device: libstk's tiled transpose (stk_transpose); host
    tensors (the CPU tests of the communication layer) take the torch view."""
    M, ld = buf.shape
    if buf.is_cuda:
        src = src.to(buf.device).contiguous()
        _lib.transpose(src, n_loc, M, M, buf, ld, zero_to=ld)
    else:
        buf[:, :n_loc].copy_(src.t())


def _time_major(buf, n_loc):
    """Contiguous (n_loc, M) block X_loc[t][i] of a slab."""
    M, ld = buf.shape
    if not buf.is_cuda:
        return buf[:, :n_loc].t().contiguous()
    out = torch.empty((n_loc, M), dtype=torch.float64, device=buf.device)
    _lib.transpose(buf, M, n_loc, ld, out, M)
    return out


def _time_rows(buf, t_idx):
    """(len(t_idx), M) contiguous copies of the time rows t_idx of a slab: ONE
    pass over the slab's lines (stk_halo_pack / stk_slab_extract_time_rows)
    instead of a strided copy per row."""
    M, ld = buf.shape
    if not buf.is_cuda:
        return torch.stack([buf[:, t] for t in t_idx]).contiguous()
    out = torch.empty((len(t_idx), M), dtype=torch.float64, device=buf.device)
    idx = _lib.to_dev(np.asarray(t_idx, dtype=np.int32))
    _lib.check(_lib.lib().stk_slab_extract_time_rows(
        _lib.stream(), M, len(t_idx), _lib.ptr(idx), _lib.ptr(buf), ld,
        _lib.ptr(out), M))
    return out


def halo_route_plan(size, M, routes):
    """Who sends which piece of which halo row to whom, in two phases, when every
    row is cut into `routes` pieces that travel over different links: piece 0 goes
    straight to the neighbour, piece j >= 1 through an intermediate rank (two hops).
    A time slab sends its whole boundary row to ONE neighbour, i.e. over one of the
    seven xGMI links of its GPU; on a fully connected node the other six idle.

    Returns a list of transfers (phase, sender, receiver, row owner, direction,
    a, b): `direction` +1 = the owner's LAST time row on its way to rank owner+1
    (that rank's X_lo), -1 = its FIRST row on its way to owner-1 (X_hi); [a, b) =
    the piece.  The list is the same on every rank and ordered, so that the
    messages between any two ranks are posted in the same order on both sides."""
    routes = max(1, min(int(routes), size - 1))
    cuts = np.linspace(0, M, routes + 1).astype(np.int64)
    plan = []
    for phase in (1, 2):
        for owner in range(size):
            for direction in (+1, -1):
                dest = owner + direction
                if not 0 <= dest < size:
                    continue
                others = [q for q in range(size) if q not in (owner, dest)]
                for j in range(routes):
                    a, b = int(cuts[j]), int(cuts[j + 1])
                    if b <= a:
                        continue
                    if j == 0 or not others:
                        if phase == 1:
                            plan.append((1, owner, dest, owner, direction, a, b))
                        continue
                    via = others[(j - 1 + owner) % len(others)]
                    if phase == 1:
                        plan.append((1, owner, via, owner, direction, a, b))
                    else:
                        plan.append((2, via, dest, owner, direction, a, b))
    return plan


def probe_halo_form(dofs_distr, candidates=(3, 7), reps=3):
    """Start-up choice of the halo form (communicate_bdr, reference
    mpi_vector.py:140-187): one REAL boundary row per neighbour is exchanged
    directly and routed over k = 3, 7 links (halo_route_plan); a routed form is
    taken only if every rank received bit for bit what the direct exchange
    delivered AND the slowest rank's time over `reps` repetitions beats the direct
    form's.  Otherwise the direct exchange stays.  Collective: every rank must call
    it; the decision is taken on reduced values, so all ranks agree.  Sets
    KronVectorMPI.HALO_ROUTES and returns the record
    {'chosen', 'ms': {k: max-over-ranks ms}, 'identical': {k: bool}, 'reason'}."""
    import torch.distributed as dist
    comm, size = dofs_distr.comm, dofs_distr.size
    rec = {'chosen': 1, 'ms': {}, 'identical': {}, 'reason': ''}
    if size <= 2:
        rec['reason'] = 'fewer than 3 ranks: no intermediate rank to route through'
        KronVectorMPI.HALO_ROUTES = 1
        return rec
    rng = np.random.RandomState(4321 + dofs_distr.rank)
    x = KronVectorMPI(dofs_distr, rng.rand(dofs_distr.t_end - dofs_distr.t_begin, dofs_distr.M))

    def sync():
        if x.buf.is_cuda:
            torch.cuda.synchronize()

    def reduce(value, op):
        t = torch.tensor([value], dtype=torch.float64, device=comm._device())
        dist.all_reduce(t, op=op, group=comm.group)
        return float(t.item())

    def run(k):
        KronVectorMPI.HALO_ROUTES = k
        x._invalidate()
        x.communicate_bdr()  # untimed first exchange: buffers, connections
        sync()
        got = x.ghost_pair().clone()
        worst = 0.0
        for _ in range(reps):
            x._invalidate()
            comm.Barrier()
            t0 = MPI.Wtime()
            x.communicate_bdr()
            sync()
            worst = max(worst, MPI.Wtime() - t0)
        return got, reduce(worst, dist.ReduceOp.MAX) * 1e3

    saved = KronVectorMPI.HALO_ROUTES
    try:
        want, rec['ms'][1] = run(1)
        for k in candidates:
            if k > size - 1:
                continue
            got, ms = run(k)
            same = reduce(1.0 if torch.equal(got, want) else 0.0, dist.ReduceOp.MIN) == 1.0
            rec['ms'][k], rec['identical'][k] = ms, same
    except Exception:
        KronVectorMPI.HALO_ROUTES = saved
        raise
    good = [k for k, same in rec['identical'].items() if same and rec['ms'][k] < rec['ms'][1]]
    rec['chosen'] = min(good, key=lambda k: rec['ms'][k]) if good else 1
    rec['reason'] = ('routed over %d links: identical rows, %.3f ms against %.3f ms direct'
                     % (rec['chosen'], rec['ms'][rec['chosen']], rec['ms'][1]) if good else
                     'direct: no routed form was both identical and faster')
    KronVectorMPI.HALO_ROUTES = rec['chosen']
    return rec


def choose_halo_form(dofs_distr):
    """The halo form of a multi-rank run, decided once at start-up (bench.py and the
    driver call this; collective).  STK_HALO_ROUTES=k pins k routes; =auto runs
    probe_halo_form.  UNSET: the direct exchange on RCCL -- the routed two-phase
    point-to-point pattern has only ever run over gloo, a stall of it on a real node
    would hang the start-up of every multi-GPU run and no `except` catches a hang
     -- and the probe on gloo, where the tests exercise it.  Once a node has
    passed `STK_HALO_ROUTES=auto`, that can become the default."""
    import os
    import torch.distributed as dist
    pinned = os.environ.get('STK_HALO_ROUTES')
    if pinned is None:
        on_gloo = dofs_distr.size > 1 and dist.is_initialized() and dist.get_backend(
            dofs_distr.comm.group) == 'gloo'
        pinned = 'auto' if on_gloo else '1'
        why = 'STK_HALO_ROUTES unset: %s' % ('probe (gloo)' if on_gloo else 'direct on RCCL (probe is opt-in: =auto)')
    else:
        why = 'STK_HALO_ROUTES=%s' % pinned
    if pinned == 'auto':
        rec = probe_halo_form(dofs_distr)
        rec['policy'] = why
        return rec
    KronVectorMPI.HALO_ROUTES = int(pinned)
    return {'chosen': int(pinned), 'reason': 'not probed', 'policy': why}


def startup_report(dofs_distr, tensors=()):
    """First-contact checks of a multi-rank run, on stderr: every rank asserts that
    the given plan tensors live on ITS device (LOCAL_RANK), and reports the
    backend, the RCCL version and which peers its GPU can access directly."""
    import os
    import sys
    import torch.distributed as dist
    comm = dofs_distr.comm
    dev = _lib.compute_device()
    for t in tensors:
        if t is not None and t.is_cuda:
            assert t.device == dev, 'plan tensor on %s, this rank computes on %s' % (t.device, dev)
    info = {'rank': comm.rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device': str(dev),
            'backend': dist.get_backend(comm.group) if comm.distributed else 'none'}
    if dev.type == 'cuda':
        assert dev.index == info['local_rank'] % torch.cuda.device_count(), info
        info['peer_access'] = [bool(p == dev.index or torch.cuda.can_device_access_peer(dev.index, p))
                               for p in range(torch.cuda.device_count())]
        try:
            info['rccl'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:  # no RCCL in this build of PyTorch
            info['rccl'] = 'unavailable (%s)' % type(exc).__name__
    print('stk start-up: %s' % info, file=sys.stderr, flush=True)
    return info


_dot_ws = {}


def _dot_workspace(device, M, n_loc, N, t_begin):
    """(scratch of stk_slab_dot, the N per-time-step sums) for this slab of the time
    axis (t_begin in the key: ranks that live in one process, as threads in
    tests/thread_comm.py, must not share the buffers)."""
    key = (device, M, n_loc, N, t_begin, threading.get_ident())
    ws = _dot_ws.get(key)
    if ws is None:
        n = int(_lib.lib().stk_slab_dot_work_size(M, n_loc))
        ws = (torch.empty(n, dtype=torch.float64, device=device),
              torch.zeros(N, dtype=torch.float64, device=device))
        _dot_ws[key] = ws
    return ws


class KronVectorMPI:
    """A vector distributed in its first (time) component
    (reference mpi_vector.py:41-240)."""
    # NumPy scalars must defer to __rmul__ instead of broadcasting over us
    __array_ufunc__ = None

    def __init__(self, dofs_distr, initial_data=None):
        self._describe(dofs_distr)
        self.reset(initial_data)

    @classmethod
    def around(cls, dofs_distr, buf):
        """The vector whose slab IS `buf` (an (M, ld) float64 device tensor an operator
        just produced; ld = the local time steps rounded up to even): no copy."""
        out = cls.__new__(cls)
        out._describe(dofs_distr)
        assert tuple(buf.shape) == (out.M, out.ld) and buf.dtype == torch.float64, (tuple(buf.shape), out.M, out.ld)
        out.communicated_bdr = False
        out.X_lo = out.X_hi = out._ghost = out._ghost_il = out._halo_send = out._records = None
        out._buf = buf
        return out

    def _describe(self, dofs_distr):
        self.dofs_distr = dofs_distr

        # Convenience
        self.t_begin = dofs_distr.t_begin
        self.t_end = dofs_distr.t_end
        self.N = dofs_distr.N
        self.M = dofs_distr.M
        self.rank = dofs_distr.rank
        self.n_loc = self.t_end - self.t_begin
        # even leading dimension: time pairs are 16-byte aligned (include/stk.h)
        # (rows padded to 128 bytes -- ld = 16 for the 9-step slabs of an 8-rank run --
        # were measured and lose: S 3.05 -> 4.44 ms, P 5.69 -> 7.21 ms,
        # profiles/r04_b_op_J3_J9_ld16.log; the bytes count, not the alignment)
        self.ld = self.n_loc + (self.n_loc & 1)
        self._pending = None

    # -- storage -------------------------------------------------------------
    @property
    def buf(self):
        """Device slab, shape (M, ld), float64."""
        return self._buf

    @property
    def X_loc(self):
        """(N_loc, M) view in the reference's orientation."""
        return self.buf[:, :self.n_loc].t()

    def reset(self, initial_data=None):
        self.communicated_bdr = False
        # ghost time rows (the reference's X_loc_bdr[0] and X_loc_bdr[-1])
        self.X_lo = self.X_hi = self._ghost = self._ghost_il = self._halo_send = self._records = None
        self._notify_pending()
        dev = _lib.compute_device()
        if initial_data is None:
            self._buf = torch.zeros((self.M, self.ld),
                                    dtype=torch.float64,
                                    device=dev)
        else:
            assert tuple(initial_data.shape) == (self.n_loc, self.M)
            self._buf = torch.zeros((self.M, self.ld),
                                    dtype=torch.float64,
                                    device=dev)
            src = initial_data if torch.is_tensor(
                initial_data) else torch.from_numpy(
                    np.ascontiguousarray(initial_data, dtype=np.float64))
            _fill_slab(self._buf, self.n_loc, src)

    def copy(self):
        cpy = KronVectorMPI.__new__(KronVectorMPI)
        cpy.__dict__.update(self.__dict__)
        cpy._pending = None
        cpy.communicated_bdr = False
        cpy.X_lo = cpy.X_hi = cpy._ghost = cpy._ghost_il = cpy._halo_send = cpy._records = None
        cpy._buf = self.buf.clone()
        return cpy

    def _like(self):
        out = KronVectorMPI.__new__(KronVectorMPI)
        out.__dict__.update(self.__dict__)
        out._pending = None
        out.communicated_bdr = False
        out.X_lo = out.X_hi = out._ghost = out._ghost_il = out._halo_send = out._records = None
        out._buf = torch.empty_like(self.buf)
        return out

    def _notify_pending(self):
        # lazy `alpha * self` expressions must see the value from before the
        # mutation that is about to happen
        if getattr(self, '_pending', None):
            for s in list(self._pending):
                s._force()
            self._pending = None

    def _invalidate(self):
        """Call before mutating: drops the cached ghost rows
        (reference mpi_vector.py:77-82)."""
        self.communicated_bdr = False
        self._notify_pending()

    # -- arithmetic (libstk BLAS-1) -------------------------------------------
    def _axpby(self, a, x_buf, b):
        n = self.buf.numel()
        _lib.check(_lib.lib().stk_axpby(_lib.stream(), n, a, _lib.ptr(x_buf),
                                        b, _lib.ptr(self.buf)))

    def __iadd__(self, other):
        self._invalidate()
        if isinstance(other, _ScaledVector) and other._lazy:
            self._axpby(other._alpha, other._src.buf, 1.0)
        else:
            self._axpby(1.0, other.buf, 1.0)
        return self

    def __isub__(self, other):
        self._invalidate()
        if isinstance(other, _ScaledVector) and other._lazy:
            self._axpby(-other._alpha, other._src.buf, 1.0)
        else:
            self._axpby(-1.0, other.buf, 1.0)
        return self

    def __imul__(self, other):
        self._invalidate()
        self._axpby(float(other), self.buf, 0.0)
        return self

    def scale_add(self, factor, other):
        """self = factor * self + other in ONE pass, bit for bit the two steps
        `self *= factor; self += other` (stk_axpby rounds factor * self before its
        fused multiply-add with 1.0 * other): PCG's update of the search direction
        (reference linalg.py:39-40) without the extra read and write of the slab."""
        self._invalidate()
        self._axpby(1.0, other.buf, float(factor))
        return self

    def __itruediv__(self, other):
        self._invalidate()
        self._axpby(1.0 / float(other), self.buf, 0.0)
        return self

    def _combine(self, a, other, b):
        """a * self + b * other as a new vector (one pass)."""
        xa, xb = self, other
        if isinstance(xa, _ScaledVector) and xa._lazy:
            a, xa = a * xa._alpha, xa._src
        if isinstance(xb, _ScaledVector) and xb._lazy:
            b, xb = b * xb._alpha, xb._src
        out = xa._like()
        _lib.check(_lib.lib().stk_axpbyz(_lib.stream(), out.buf.numel(), a,
                                         _lib.ptr(xa.buf), b,
                                         _lib.ptr(xb.buf), _lib.ptr(out.buf)))
        return out

    def __add__(self, other):
        return self._combine(1.0, other, 1.0)

    def __sub__(self, other):
        return self._combine(1.0, other, -1.0)

    def __rmul__(self, other):
        return _ScaledVector(float(other), self)

    def __mul__(self, other):
        return _ScaledVector(float(other), self)

    def __neg__(self):
        return _ScaledVector(-1.0, self)

    def __truediv__(self, other):
        return _ScaledVector(1.0 / float(other), self)

    def dot(self, vec_other):
        """Global inner product (reference mpi_vector.py:205-210).  The reference
        adds the products slab by slab and all-reduces one number, so its value
        depends on the number of ranks in the last digits.  Here every TIME STEP is
        summed over the spatial index in a fixed shape (stk_slab_dot), the N
        per-step sums are all-reduced (each has one contributor: exact) and added in
        increasing t on every rank: one value whatever the partition of the time
        axis, bit for bit; one D2H read of N doubles."""
        assert (isinstance(vec_other, KronVectorMPI))
        assert (vec_other.buf.shape == self.buf.shape)
        work, steps = _dot_workspace(self.buf.device, self.M, self.n_loc, self.N, self.t_begin)
        _lib.check(_lib.lib().stk_slab_dot(
            _lib.stream(), self.M, self.n_loc, self.ld, _lib.ptr(self.buf),
            _lib.ptr(vec_other.buf), _lib.ptr(work), self.N, self.t_begin,
            _lib.ptr(steps)))
        self.dofs_distr.comm.allreduce_tensor_(steps)
        total = 0.0
        for value in steps.tolist():  # increasing t, plain additions (stk_sum_steps)
            total += value
        return total

    # -- I/O -------------------------------------------------------------------
    def scatter(self, X_glob):
        """Root's flat global array -> slabs (reference mpi_vector.py:124-132)."""
        comm, dd = self.dofs_distr.comm, self.dofs_distr
        self._invalidate()
        dev = self.buf.device
        if comm.size == 1:
            slab = torch.from_numpy(
                np.ascontiguousarray(X_glob, dtype=np.float64).reshape(
                    self.N, self.M))
        else:
            slab = torch.empty((self.n_loc, self.M), dtype=torch.float64)
            if comm.rank == 0:
                X = torch.from_numpy(
                    np.ascontiguousarray(X_glob, dtype=np.float64).reshape(
                        self.N, self.M))
                sends = [(X[b:e].contiguous().to(comm._device()), p)
                         for p, (b, e) in enumerate(dd.dof_distribution)
                         if p != 0]
                reqs = comm.exchange(sends, [])
                slab = X[dd.dof_distribution[0][0]:dd.dof_distribution[0][1]]
                comm.wait_all(reqs)
            else:
                slab = slab.to(comm._device())
                comm.wait_all(comm.exchange([], [(slab, 0)]))
        _fill_slab(self._buf, self.n_loc, slab)

    def gather(self, X_glob):
        """Slabs -> root's flat global array (reference mpi_vector.py:134-138)."""
        comm, dd = self.dofs_distr.comm, self.dofs_distr
        mine = _time_major(self.buf, self.n_loc)
        if comm.size == 1:
            X_glob[...] = mine.cpu().numpy().reshape(X_glob.shape)
            return
        if comm.rank == 0:
            out = np.asarray(X_glob).reshape(self.N, self.M)
            bufs = {
                p: torch.empty((e - b, self.M),
                               dtype=torch.float64,
                               device=comm._device())
                for p, (b, e) in enumerate(dd.dof_distribution) if p != 0
            }
            reqs = comm.exchange([], [(t, p) for p, t in bufs.items()])
            out[self.t_begin:self.t_end] = mine.cpu().numpy()
            comm.wait_all(reqs)
            for p, t in bufs.items():
                b, e = dd.dof_distribution[p]
                out[b:e] = t.cpu().numpy()
        else:
            comm.wait_all(comm.exchange([(mine.to(comm._device()), 0)], []))

    # -- communication -----------------------------------------------------------
    def communicate_bdr(self, callback=None, records=False):
        """Fetches the ghost time rows t_begin-1 and t_end from the neighbour
        ranks into X_lo / X_hi while `callback` computes what does not need
        them; cached until the vector is mutated
        (reference mpi_vector.py:140-187)."""
        if self.communicated_bdr:
            if callback is not None:
                callback()
            return 0.0
        comm = self.dofs_distr.comm
        rank, size = self.rank, self.dofs_distr.size
        sends, recvs = [], []
        if size > 1 and getattr(self, '_ghost', None) is None:
            # row 0 = X_lo (time row t_begin - 1), row 1 = X_hi (row t_end)
            self._ghost = torch.zeros((2, self.M),
                                      dtype=torch.float64,
                                      device=self.buf.device)
        if size > 1:
            # the first and the last time step, both from one pass over the slab
            # (stk_halo_pack) into a reused (2, M) send buffer
            if getattr(self, '_halo_send', None) is None:
                self._halo_send = torch.empty((2, self.M), dtype=torch.float64,
                                              device=self.buf.device)
            first = self._halo_send[0] if rank > 0 else None
            last = self._halo_send[1] if rank + 1 < size else None
            if self.buf.is_cuda and records:
                # ... and, from the same pass, per spatial dof the four entries the boundary
                # steps of the overlapped Kronecker apply read (boundary_records)
                if getattr(self, '_records', None) is None:
                    self._records = torch.empty((self.M, 4), dtype=torch.float64,
                                                device=self.buf.device)
                _lib.check(_lib.lib().stk_halo_pack_records(
                    _lib.stream(), self.M, self.n_loc, self.ld, _lib.ptr(self.buf),
                    _lib.ptr(first), 1, _lib.ptr(last), 1, _lib.ptr(self._records)))
            elif self.buf.is_cuda:
                _lib.check(_lib.lib().stk_halo_pack(
                    _lib.stream(), self.M, self.n_loc, self.ld, _lib.ptr(self.buf),
                    _lib.ptr(first), 1, _lib.ptr(last), 1))
            else:
                if first is not None:
                    first.copy_(self.buf[:, 0])
                if last is not None:
                    last.copy_(self.buf[:, self.n_loc - 1])
        if rank > 0:
            self.X_lo = self._ghost[0]
        if rank + 1 < size:
            self.X_hi = self._ghost[1]
        if self.HALO_ROUTES > 1 and size > 2:
            return self._routed_halo(first, last, callback)
        if rank > 0:
            sends.append((first, rank - 1))
            recvs.append((self.X_lo, rank - 1))
        if rank + 1 < size:
            sends.append((last, rank + 1))
            recvs.append((self.X_hi, rank + 1))
        reqs = comm.exchange(sends, recvs)

        # Do computation that doesn't require the bdr to be present.
        if callback is not None:
            callback()

        start_time = MPI.Wtime()
        comm.wait_all(reqs)
        time_communication = MPI.Wtime() - start_time
        self.communicated_bdr = True
        self._ghost_il_stale = True
        return time_communication

    # Pieces a halo row is cut into so that they travel over different links
    # (environment STK_HALO_ROUTES; 1 = the whole row straight to the neighbour).
    # Opt-in on RCCL: it has only ever run over gloo -- no multi-GPU node was available
    # to measure it or to prove it there.  STK_HALO_ROUTES=auto:
    # probe_halo_form decides at start-up; unset: choose_halo_form (direct on RCCL, the
    # probe on gloo).
    HALO_ROUTES = int(__import__('os').environ.get('STK_HALO_ROUTES', '1').replace('auto', '1'))

    def _routed_halo(self, first, last, callback):
        """communicate_bdr with every row cut into HALO_ROUTES pieces: phase 1 =
        the direct pieces and the first hops, phase 2 = the intermediates forward
        what they hold (halo_route_plan)."""
        comm, rank, size = self.dofs_distr.comm, self.rank, self.dofs_distr.size
        plan = halo_route_plan(size, self.M, self.HALO_ROUTES)
        held = {}

        def source(owner, direction, a, b):  # a piece of MY row
            return (last if direction > 0 else first)[a:b]

        def target(direction, a, b):  # where a piece of a neighbour's row ends up
            return (self.X_lo if direction > 0 else self.X_hi)[a:b]

        waited = 0.0
        for phase in (1, 2):
            sends, recvs = [], []
            for ph, src, dst, owner, direction, a, b in plan:
                if ph != phase:
                    continue
                final = dst == owner + direction
                if src == rank:
                    buf = source(owner, direction, a, b) if owner == rank else held[(owner, direction, a)]
                    sends.append((buf, dst))
                if dst == rank:
                    if final:
                        recvs.append((target(direction, a, b), src))
                    else:
                        held[(owner, direction, a)] = torch.empty(
                            b - a, dtype=torch.float64, device=self.buf.device)
                        recvs.append((held[(owner, direction, a)], src))
            reqs = comm.exchange(sends, recvs)
            if phase == 1 and callback is not None:
                callback()  # what does not need the halo, beside the first phase
            start_time = MPI.Wtime()
            comm.wait_all(reqs)
            waited += MPI.Wtime() - start_time
        self.communicated_bdr = True
        self._ghost_il_stale = True
        return waited

    def boundary_records(self):
        """(M, 4) rows (x[j][0], x[j][1], x[j][n_loc-2], x[j][n_loc-1]) left by
        communicate_bdr(records=True) -- valid until the vector changes."""
        assert self.communicated_bdr and getattr(self, '_records', None) is not None
        return self._records

    def ghost_pair(self):
        """(2, M) buffer [X_lo; X_hi] filled by communicate_bdr."""
        return self._ghost

    def ghost_interleaved(self):
        """(M, 2) buffer with row j = (X_lo[j], X_hi[j]) -- the layout in which
        stk_kron_pack_apply gathers both ghost time steps with one 16-byte
        load; a side without a neighbour is zero.  Valid after
        communicate_bdr, rebuilt when the halo is."""
        assert self.communicated_bdr
        if getattr(self, '_ghost_il', None) is None or self._ghost_il_stale:
            if getattr(self, '_ghost_il', None) is None:
                self._ghost_il = torch.empty((self.M, 2), dtype=torch.float64,
                                             device=self.buf.device)
            _lib.check(_lib.lib().stk_interleave_ghosts(
                _lib.stream(), self.M, _lib.ptr(self.X_lo), _lib.ptr(self.X_hi),
                _lib.ptr(self._ghost_il)))
            self._ghost_il_stale = False
        return self._ghost_il

    def communicate_dofs(self, comm_dofs):
        """Fetches arbitrary remote time rows.  `comm_dofs` = (local row,
        remote row) pairs of a time matrix with symmetric sparsity pattern
        (reference mpi_vector.py:189-203).  Each needed row travels once per
        destination; messages between a pair of ranks are ordered by
        (row sent, rank) on both sides, which replaces MPI tags.
        Returns (recv buffer (n_recv, M), {remote row: slot}, requests)."""
        dd = self.dofs_distr
        need = sorted(set(int(r) for _, r in comm_dofs))
        slot = {r: k for k, r in enumerate(need)}
        recv_buf = torch.empty((len(need), self.M),
                               dtype=torch.float64,
                               device=self.buf.device)
        send_set = sorted(
            set((int(s), int(dd.dof2proc[int(r)])) for s, r in comm_dofs))
        rows = sorted(set(s - self.t_begin for s, _ in send_set))
        packed = _time_rows(self.buf, rows) if rows else None
        sends = [(packed[rows.index(s - self.t_begin)], p) for s, p in send_set]
        recvs = [(recv_buf[slot[r]], int(dd.dof2proc[r])) for r in need]
        reqs = dd.comm.exchange(sends, recvs)
        return recv_buf, slot, reqs

    def permute(self, vec_perm=None):
        """Swaps the roles of time and space: all-to-all transpose
        (reference mpi_vector.py:212-240).  Returns (vector distributed over
        the M space dofs with N as second component, communication time)."""
        start_time = MPI.Wtime()
        comm = self.dofs_distr.comm
        if vec_perm is None:
            vec_perm = KronVectorMPI(DofDistributionMPI(comm, self.M, self.N))
        else:
            assert (vec_perm.N == self.M and vec_perm.M == self.N)
            vec_perm._invalidate()
        pd = vec_perm.dofs_distr
        x_begin, x_end = vec_perm.t_begin, vec_perm.t_end
        # own block needs no transfer
        ld, ldp = self.ld, vec_perm.ld

        def own_block(xb, xe, tb):
            # vec_perm.buf[tb + t, x - xb] = self.buf[x, t]  (stk_transpose)
            if self.buf.is_cuda:
                _lib.transpose(self.buf, xe - xb, self.n_loc, ld, vec_perm._buf,
                               ldp, src_off=xb * ld, dst_off=tb * ldp)
            else:
                vec_perm._buf[tb:tb + self.n_loc, :xe - xb].copy_(
                    self.buf[xb:xe, :self.n_loc].t())

        if comm.size == 1:
            own_block(x_begin, x_end, 0)
            return vec_perm, MPI.Wtime() - start_time
        sends, recvs, staged = [], [], []
        for p in range(comm.size):
            xb, xe = pd.dof_distribution[p]
            tb, te = self.dofs_distr.dof_distribution[p]
            if p == comm.rank:
                own_block(xb, xe, tb)
                continue
            # the block travels already transposed: (n_loc, xe - xb), time-major
            sbuf = torch.empty((self.n_loc, xe - xb), dtype=torch.float64,
                               device=self.buf.device)
            if xe > xb:
                if self.buf.is_cuda:
                    _lib.transpose(self.buf, xe - xb, self.n_loc, ld, sbuf,
                                   xe - xb, src_off=xb * ld)
                else:
                    sbuf.copy_(self.buf[xb:xe, :self.n_loc].t())
            sends.append((sbuf, p))
            rbuf = torch.empty((te - tb, x_end - x_begin),
                               dtype=torch.float64,
                               device=self.buf.device)
            recvs.append((rbuf, p))
            staged.append((tb, te, rbuf))
        comm.wait_all(comm.exchange(sends, recvs))
        nx = x_end - x_begin
        for tb, te, rbuf in staged:
            if rbuf.is_cuda:
                _lib.copy_block(rbuf, te - tb, nx, nx, vec_perm._buf, ldp,
                                dst_off=tb * ldp)
            else:
                vec_perm._buf[tb:te, :nx].copy_(rbuf)
        return vec_perm, MPI.Wtime() - start_time


class _ScaledVector(KronVectorMPI):
    """`alpha * x`, evaluated lazily: consumed by += / -= / + / - as a fused
    axpy, or materialised on first use as an ordinary vector."""
    def __init__(self, alpha, src):
        if isinstance(src, _ScaledVector) and src._lazy:
            alpha, src = alpha * src._alpha, src._src
        self.__dict__.update(src.__dict__)
        self._pending = None
        self.communicated_bdr = False
        self.X_lo = self.X_hi = self._ghost = self._halo_send = self._records = None
        self._alpha, self._src, self._lazy = alpha, src, True
        self.__dict__.pop('_buf', None)
        if src._pending is None:
            src._pending = weakref.WeakSet()
        src._pending.add(self)

    def _force(self):
        if self._lazy:
            src = self._src
            buf = torch.empty_like(src.buf)
            _lib.check(_lib.lib().stk_axpbyz(_lib.stream(), buf.numel(),
                                             self._alpha, _lib.ptr(src.buf),
                                             0.0, None, _lib.ptr(buf)))
            self.__dict__['_buf'] = buf
            self._lazy = False
            if src._pending is not None:
                src._pending.discard(self)
            self._src = None

    @property
    def _buf(self):
        self._force()
        return self.__dict__['_buf']

    @_buf.setter
    def _buf(self, value):
        self._lazy = False
        self.__dict__['_buf'] = value
