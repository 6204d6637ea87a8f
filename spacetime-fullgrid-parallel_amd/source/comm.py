"""A small MPI-like facade over torch.distributed.

The reference talks to mpi4py (``from mpi4py import MPI``; every call site is
listed in SURVEY.md section 2.2).  Here one process drives one GPU and the
transport is torch.distributed: backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" on CPU for the world_size-2 tests.  Only what the hot path uses exists:
rank/size, scalar allreduce, neighbour / partner point-to-point exchange,
barrier, and object bcast/gather for driver bookkeeping.

Usage mirrors the reference:  ``from source.comm import MPI`` then
``MPI.COMM_WORLD``, ``MPI.Wtime()``.
"""
import os
import time

import torch
import torch.distributed as dist


class Comm:
    """Communicator over a torch.distributed process group (None = a single
    process without torch.distributed).

    A one-rank group normally short-cuts every collective.  With
    STK_FORCE_COLLECTIVES=1 it does not: the scalar all-reduce, barrier and
    object broadcast / gather then run on the backend (RCCL) even at world size
    1 -- the transport can be executed on a one-GPU box."""
    # Set (the drivers and bench.py's per-rank pass do): every scalar all-reduce is
    # bracketed by host time stamps after the device has drained, so that
    # `allreduce_host_s` is the latency of the collective itself (plus the read of
    # its result), not the wait for the kernels before it.
    timing = False

    def __init__(self, group=None, distributed=None):
        self.allreduce_calls = 0
        self.allreduce_host_s = 0.0
        self.group = group
        self.distributed = dist.is_initialized() if distributed is None else distributed
        if self.distributed:
            self.rank = dist.get_rank(group)
            self.size = dist.get_world_size(group)
        else:
            self.rank, self.size = 0, 1
        self.collective = self.size > 1 or (
            self.distributed and os.environ.get('STK_FORCE_COLLECTIVES') == '1')

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    # -- collectives ---------------------------------------------------------
    def allreduce(self, value):
        """Sum of a Python float over the ranks (reference mpi_vector.py:209)."""
        if not self.collective:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._device())
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    def allreduce_tensor_(self, t):
        """In-place sum of a small tensor that already lives on the compute
        device (keeps the dot result on the GPU until the single D2H read)."""
        if self.collective:
            self.allreduce_calls += 1
            began = None
            if Comm.timing:
                if t.is_cuda:
                    torch.cuda.synchronize()
                began = time.perf_counter()
            if t.is_cuda and self._device().type == 'cpu':
                host = t.cpu()  # gloo: stage through the host
                dist.all_reduce(host, group=self.group)
                t.copy_(host)
            else:
                dist.all_reduce(t, group=self.group)
            if began is not None:
                if t.is_cuda:
                    torch.cuda.synchronize()
                self.allreduce_host_s += time.perf_counter() - began
        return t

    def reset_counters(self):
        self.allreduce_calls, self.allreduce_host_s = 0, 0.0

    def bcast(self, obj, root=0):
        if not self.collective:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=root, group=self.group)
        return box[0]

    def gather(self, obj, root=0):
        if not self.collective:
            return [obj]
        out = [None] * self.size if self.rank == root else None
        dist.gather_object(obj, out, dst=root, group=self.group)
        return out

    def Barrier(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.collective:
            dist.barrier(group=self.group)

    # -- point to point --------------------------------------------------------
    def exchange(self, sends, recvs):
        """Posts all sends/recvs as one batch (one RCCL group call).
        sends/recvs: lists of (tensor, peer_rank).  Returns the requests; call
        ``wait_all`` on them.  Matching is by posting order per peer pair, so
        both sides must post their transfers with a common ordering (the
        callers order by global time index)."""
        if not sends and not recvs:
            return []
        staged = []
        if self._device().type == 'cpu':
            # gloo moves host memory: device tensors are staged through the
            # host (CPU tests, and several ranks sharing one GPU)
            sends = [(t.cpu() if t.is_cuda else t, p) for t, p in sends]
            host_recvs = []
            for t, p in recvs:
                if t.is_cuda:
                    h = torch.empty(t.shape, dtype=t.dtype)
                    staged.append((t, h))
                    host_recvs.append((h, p))
                else:
                    host_recvs.append((t, p))
            recvs = host_recvs
        ops = [dist.P2POp(dist.isend, t, p, self.group) for t, p in sends]
        ops += [dist.P2POp(dist.irecv, t, p, self.group) for t, p in recvs]
        reqs = list(dist.batch_isend_irecv(ops))
        if staged:
            reqs.append(_CopyBack(staged))
        return reqs

    @staticmethod
    def wait_all(reqs):
        for r in reqs:
            r.wait()

    def _device(self):
        if dist.get_backend(self.group) == 'nccl':
            from . import _lib
            return _lib.compute_device()
        return torch.device('cpu')


class _CopyBack:
    """Pseudo request: after the real receives have been waited for (it sits
    last in the list), move the staged host buffers to their device tensors."""
    def __init__(self, pairs):
        self.pairs = pairs

    def wait(self):
        for dev, host in self.pairs:
            dev.copy_(host)


def init_from_env():
    """Joins the job torchrun started (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*), one process per GPU.  A no-op for a plain single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if torch.cuda.is_available():
        # several ranks may share one GPU in tests (STK_BACKEND=gloo)
        from . import _lib
        _lib.set_process_device(
            int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    force = os.environ.get('STK_FORCE_COLLECTIVES') == '1' and 'RANK' in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        backend = os.environ.get(
            'STK_BACKEND', 'nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            from . import _lib
            kw['device_id'] = _lib.compute_device()
        dist.init_process_group(backend, **kw)
    return Comm()


class _MPI:
    """Namespace with the two names the drivers use."""
    _world = None

    @property
    def COMM_WORLD(self):
        if self._world is None or (dist.is_initialized()
                                   and not self._world.distributed):
            self._world = init_from_env()
        return self._world

    @staticmethod
    def Wtime():
        return time.perf_counter()


MPI = _MPI()
