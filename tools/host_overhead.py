import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/spacetime-fullgrid-parallel_amd')
burn = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
mode = sys.argv[2] if len(sys.argv) > 2 else 'fork'
if mode == 'sleep':
    time.sleep(burn)
elif mode == 'forkidle':
    import multiprocessing as mp
    with mp.get_context('fork').Pool(64) as pool:
        pool.map(time.sleep, [burn] * 64)
elif mode == 'child':
    import subprocess
    subprocess.run([sys.executable, '-c', 'import time; time.sleep(%f)' % burn])
elif burn > 0:  # 'fork' and 'recover'
    import multiprocessing as mp
    def spin(t):
        t0=time.time(); a=np.random.rand(512,512)
        while time.time()-t0 < t: a = a @ a; a /= np.abs(a).max()
        return 0
    with mp.get_context('fork').Pool(64) as pool:
        pool.map(spin, [burn]*64)
import torch
from source.assembly import space_matrices, time_matrices
from source.problem import problem_helper
from source.comm import MPI
from source.mpi_kron import SumMPI, TridiagKronMatMPI
from source.mpi_vector import DofDistributionMPI, KronVectorMPI
mesh_space,_,mesh_time,_,_ = problem_helper('square', J_space=9, J_time=6)
A_t,L_t,M_t,G_t,u0 = time_matrices(mesh_time); M_x,A_x = space_matrices(mesh_space, scipy_path=True)
dd = DofDistributionMPI(MPI.COMM_WORLD, A_t.shape[0], M_x.shape[0])
op = SumMPI(dd,[TridiagKronMatMPI(dd,A_t,M_x),TridiagKronMatMPI(dd,M_t,A_x)])
x = KronVectorMPI(dd, np.random.rand(dd.t_end-dd.t_begin, M_x.shape[0])); y = x._like()
def step():
    x._invalidate(); op._matvec(x,y)
for _ in range(200): step()
torch.cuda.synchronize()
pauses = [0, 0, 0] + ([15, 15, 15, 15] if mode == 'recover' else [])
for pause in pauses:
    if pause:
        time.sleep(pause)
        for _ in range(50): step()
        torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(200): step()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(mode, 'pause %2d' % pause, 'burn %.0f s: host enqueue %.1f us per step, total %.1f us per step' % (burn, (t1-t0)/200*1e6, (t2-t0)/200*1e6), flush=True)
