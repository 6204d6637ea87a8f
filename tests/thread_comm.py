"""Several RANKS of the time-slab decomposition as threads of one process.

Test infrastructure only.  The GPU boxes of the pool let at most six processes
share a card, so an 8-rank run of a BASELINE configuration cannot be rehearsed as
eight processes over gloo (tests/test_distributed.py goes up to five).  Here every
rank is a thread with a communicator of the same interface as source/comm.Comm
(rank / size, allreduce_tensor_, exchange + wait_all, bcast, gather, Barrier); all
ranks launch on the one device, the messages are device-to-device copies handed
over through queues.  Everything a rank computes -- its slab kernels, the rows it
packs for its neighbours, the ghost forms of the Kronecker kernels, the per-step
partial sums of dot, the transposes of the wavelet transform -- is what it computes
as a process; only the wire differs (which the gloo tests cover on 2-8 ranks)."""
import queue
import threading

import torch


class ThreadWorld:
    def __init__(self, size, timeout=900.0):
        self.size = size
        self.timeout = timeout
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size
        self.mail = {(s, d): queue.Queue() for s in range(size) for d in range(size)}

    def wait(self):
        self.barrier.wait(self.timeout)


class _Recv:
    def __init__(self, world, box, tensor):
        self.world, self.box, self.tensor = world, box, tensor

    def wait(self):
        self.tensor.copy_(self.box.get(timeout=self.world.timeout))


class ThreadComm:
    timing = False

    def __init__(self, world, rank):
        self.world = world
        self.rank, self.size = rank, world.size
        self.group = None
        self.distributed = False  # no torch.distributed process group behind it
        self.collective = self.size > 1
        self.allreduce_calls = 0
        self.allreduce_host_s = 0.0

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def _device(self):
        from source import _lib
        return _lib.compute_device()

    # -- collectives: every rank leaves its contribution, all read all ----------
    def _all(self, value):
        w = self.world
        w.slots[self.rank] = value
        w.wait()
        got = list(w.slots)
        w.wait()  # nobody overwrites its slot before everybody has read
        return got

    def allreduce(self, value):
        return float(sum(self._all(float(value)))) if self.collective else value

    def allreduce_tensor_(self, t):
        if self.collective:
            self.allreduce_calls += 1
            parts = self._all(t.clone())
            total = parts[0].clone()
            for p in parts[1:]:
                total += p
            t.copy_(total)
        return t

    def reset_counters(self):
        self.allreduce_calls, self.allreduce_host_s = 0, 0.0

    def bcast(self, obj, root=0):
        return self._all(obj)[root] if self.collective else obj

    def gather(self, obj, root=0):
        got = self._all(obj) if self.collective else [obj]
        return got if self.rank == root else None

    def Barrier(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.collective:
            self.world.wait()

    # -- point to point: messages of a pair of ranks match in posting order -------
    def exchange(self, sends, recvs):
        for t, peer in sends:
            self.world.mail[(self.rank, peer)].put(t.clone())
        return [_Recv(self.world, self.world.mail[(peer, self.rank)], t) for t, peer in recvs]

    @staticmethod
    def wait_all(reqs):
        for r in reqs:
            r.wait()


def run_ranks(size, fn, timeout=900.0):
    """fn(comm) on `size` threads, one communicator each; returns the list of results
    (rank order) and re-raises the first failure."""
    world = ThreadWorld(size, timeout)
    out, err = [None] * size, [None] * size

    def body(r):
        try:
            out[r] = fn(ThreadComm(world, r))
        except BaseException as e:  # a failed rank must not leave the others waiting
            err[r] = e
            world.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real:
        raise real[0]
    if any(e is not None for e in err) or any(t.is_alive() for t in threads):
        raise RuntimeError('a rank thread failed or timed out: %r' % (err,))
    return out
