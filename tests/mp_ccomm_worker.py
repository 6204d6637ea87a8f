"""Worker of test_c_abi_communication: the stk_comm_* entry points of libstk (RCCL
loaded by the library itself), driven through ctypes as a C host would -- no
torch.distributed.  RANK / WORLD_SIZE come from the environment; the unique id
travels through a file, the way any launcher could hand it over.  torch is used
only to hold device buffers and to read them back."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))
from source import _lib  # noqa: E402

rank, size = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
id_path = os.environ['STK_COMM_ID_FILE']
_lib.set_process_device(int(os.environ.get('LOCAL_RANK', rank)) % torch.cuda.device_count())
lib, st = _lib.lib(), _lib.stream()
uid = (ctypes.c_char * 128)()
if rank == 0:
    _lib.check(lib.stk_comm_unique_id(uid))
    with open(id_path + '.tmp', 'wb') as f:
        f.write(bytes(uid))
    os.rename(id_path + '.tmp', id_path)
else:
    for _ in range(600):
        if os.path.exists(id_path):
            break
        time.sleep(0.1)
    uid = (ctypes.c_char * 128).from_buffer_copy(open(id_path, 'rb').read())
comm = ctypes.c_void_p()
_lib.check(lib.stk_comm_create(rank, size, uid, ctypes.byref(comm)))
dev = _lib.compute_device()

# dot's allreduce (mpi_vector.py:209)
v = torch.tensor([rank + 1.0, 2.0, -0.5 * rank], dtype=torch.float64, device=dev)
_lib.check(lib.stk_comm_allreduce_sum(comm, st, _lib.ptr(v), 3))
want = np.array([size * (size + 1) / 2.0, 2.0 * size, -0.5 * size * (size - 1) / 2.0])
assert np.array_equal(v.cpu().numpy(), want), (v, want)

# communicate_bdr (mpi_vector.py:140-187): slab -> stk_halo_pack -> exchange
M, n_loc = 1000, 5 + rank
ld = lib.stk_slab_ld(n_loc)


def slab_of(r):  # rank r's slab content, known to everybody
    return np.arange(M * (5 + r), dtype=np.float64).reshape(5 + r, M) + 1000.0 * r


x = torch.zeros((M, ld), dtype=torch.float64, device=dev)
mine = slab_of(rank)
_lib.check(lib.stk_slab_upload(st, M, n_loc, ld, mine.ctypes.data, _lib.ptr(x)))
send = torch.zeros((2, M), dtype=torch.float64, device=dev)
recv = torch.full((2, M), -1.0, dtype=torch.float64, device=dev)
_lib.check(lib.stk_halo_pack(st, M, n_loc, ld, _lib.ptr(x), _lib.ptr(send[0]), 1, _lib.ptr(send[1]), 1))
_lib.check(lib.stk_comm_halo_exchange(comm, st, M, _lib.ptr(send[0]), _lib.ptr(send[1]), _lib.ptr(recv[0]),
                                      _lib.ptr(recv[1])))
got = recv.cpu().numpy()
if rank > 0:
    assert np.array_equal(got[0], slab_of(rank - 1)[-1])
else:
    assert (got[0] == -1.0).all()
if rank + 1 < size:
    assert np.array_equal(got[1], slab_of(rank + 1)[0])
else:
    assert (got[1] == -1.0).all()

# a batch (communicate_dofs / permute): to the next rank around the ring and to myself
a = torch.arange(7, dtype=torch.float64, device=dev) + 10.0 * rank
b = torch.arange(3, dtype=torch.float64, device=dev) - 5.0 * rank
ra = torch.zeros(7, dtype=torch.float64, device=dev)
rb = torch.zeros(3, dtype=torch.float64, device=dev)
nxt, prv = (rank + 1) % size, (rank - 1) % size
sends = (_lib.CommMsg * 2)((_lib.ptr(a), 7, nxt), (_lib.ptr(b), 3, rank))
recvs = (_lib.CommMsg * 2)((_lib.ptr(ra), 7, prv), (_lib.ptr(rb), 3, rank))
if size == 1:  # both messages go to myself: keep their posting order
    sends = (_lib.CommMsg * 2)((_lib.ptr(a), 7, 0), (_lib.ptr(b), 3, 0))
    recvs = (_lib.CommMsg * 2)((_lib.ptr(ra), 7, 0), (_lib.ptr(rb), 3, 0))
_lib.check(lib.stk_comm_exchange(comm, st, 2, sends, 2, recvs))
assert np.array_equal(ra.cpu().numpy(), np.arange(7) + 10.0 * prv)
assert np.array_equal(rb.cpu().numpy(), np.arange(3) - 5.0 * rank)
# malformed requests are refused, not sent
bad = (_lib.CommMsg * 1)((_lib.ptr(a), 7, size))
assert lib.stk_comm_exchange(comm, st, 1, bad, 0, None) != 0
torch.cuda.synchronize()
_lib.check(lib.stk_comm_destroy(comm))
print('mp_ccomm_worker ok rank %d of %d' % (rank, size))
