"""Worker of tests/test_distributed.py (gpu): ONE rank under
torch.distributed.run with the backend the product ships with,
STK_BACKEND=nccl (= RCCL), and STK_FORCE_COLLECTIVES=1 so that the one-rank
group does not short-cut its collectives.  Proves on a one-GPU box what can be
proved there: process-group initialisation with a device id, device-side
all-reduce inside `dot`, barrier, object broadcast / gather (the telemetry
blob), a batched isend/irecv pair on device tensors (to the rank itself), and
the whole preconditioned solve running through that communicator, against the
CPU oracle.  The neighbour exchange between two GPUs cannot be shown here."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'spacetime-fullgrid-parallel_amd'))

import heateq_mpi as hm  # noqa: E402
from oracle.heat import HeatEquationOracle  # noqa: E402
from oracle.krylov import pcg  # noqa: E402
from source.comm import MPI  # noqa: E402
from source.linalg import PCG  # noqa: E402
from source.mpi_vector import KronVectorMPI  # noqa: E402


def main():
    assert os.environ['STK_BACKEND'] == 'nccl'
    comm = MPI.COMM_WORLD
    assert dist.is_initialized() and dist.get_backend() == 'nccl'
    assert comm.distributed and comm.collective and comm.Get_size() == 1
    assert comm._device().type == 'cuda'
    # scalar all-reduce, barrier, object collectives on RCCL
    assert comm.allreduce(2.5) == 2.5
    comm.Barrier()
    assert comm.bcast({'a': 1}) == {'a': 1}
    assert comm.gather('blob') == ['blob']
    # batched device-to-device isend/irecv (to this rank itself: the one pair a
    # one-rank group has), the call the halo exchange makes
    src = torch.arange(1000, dtype=torch.float64, device='cuda')
    dst = torch.zeros_like(src)
    comm.wait_all(comm.exchange([(src, 0)], [(dst, 0)]))
    torch.cuda.synchronize()
    assert torch.equal(src, dst)

    J_time, J_space = 3, 3
    h = hm.HeatEquationMPI(J_space=J_space, J_time=J_time)
    X = np.random.RandomState(5).rand(h.N, h.M)
    x = KronVectorMPI(h.dofs_distr, X)
    d = x.dot(x)  # device all-reduce inside
    assert abs(d - np.vdot(X, X)) < 1e-12 * d
    mats = dict(A_t=h.A_t, L_t=h.L_t, M_t=h.M_t, G_t=h.G_t, M_x=h.M_x,
                A_x=h.A_x, P_mats=h.hierarchy.P_mats, u0_t=h.u0_t, u0_x=h.u0_x)
    o = HeatEquationOracle(mats, J_time)
    hist = []
    w, its = PCG(h.WT_S_W, h.P, h.rhs, history=hist)
    wo, it_o, hist_o = pcg(o.WT_S_W, o.P, o.rhs())
    assert its == it_o, (its, it_o)
    assert np.allclose(hist, hist_o, rtol=1e-9, atol=1e-26)
    got = w.X_loc.cpu().numpy()
    assert np.linalg.norm(got - wo) < 1e-9 * np.linalg.norm(wo)
    comm.Barrier()
    print('mp_nccl_worker ok: RCCL group of size 1, %d PCG iterations' % its)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
