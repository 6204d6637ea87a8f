import sys, os, time, threading, collections
REPO='/root/repo' if os.path.isdir('/root/repo/spacetime-fullgrid-parallel_amd') else os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,'spacetime-fullgrid-parallel_amd'))
import numpy as np, torch
from source import _lib
import heateq_mpi as hm
torch.zeros(1,device='cuda')
hm.HeatEquationMPI(J_space=9,J_time=6)
orig=_lib.to_dev
stat=collections.defaultdict(lambda:[0,0.0,0]); lock=threading.Lock()
def probe(array,dtype=None):
    t=time.perf_counter(); out=orig(array,dtype); dt=time.perf_counter()-t
    nb=out.numel()*out.element_size()
    f=sys._getframe(1); key='%s:%s:%d'%(os.path.basename(f.f_code.co_filename),f.f_code.co_name,f.f_lineno)
    with lock:
        s=stat[key]; s[0]+=nb; s[1]+=dt; s[2]+=1
    return out
_lib.to_dev=probe
import source.multigrid as mg, source.linop as lo
t=time.time(); h=hm.HeatEquationMPI(J_space=9,J_time=6); torch.cuda.synchronize(); print('setup %.2f'%(time.time()-t))
tot=sum(s[0] for s in stat.values()); tt=sum(s[1] for s in stat.values()); n=sum(s[2] for s in stat.values())
print('to_dev: %d calls, %.1f MB, %.3f s of thread time (%.2f GB/s)'%(n,tot/1e6,tt,tot/1e9/max(tt,1e-9)))
for k,s in sorted(stat.items(), key=lambda kv:-kv[1][1])[:25]:
    print('%8.1f MB %7.1f ms %4d calls  %s'%(s[0]/1e6,s[1]*1e3,s[2],k))
# raw copy rates
a=np.random.rand(12_000_000)
for rep in range(2):
    t=time.perf_counter(); d=torch.from_numpy(a).to('cuda'); torch.cuda.synchronize(); print('pageable 96 MB: %.1f ms'%((time.perf_counter()-t)*1e3))
p=torch.empty(12_000_000,dtype=torch.float64).pin_memory()
t=time.perf_counter(); p.numpy()[:]=a; t1=time.perf_counter(); d=p.to('cuda',non_blocking=True); torch.cuda.synchronize(); t2=time.perf_counter()
print('copy to pinned %.1f ms, pinned H2D %.1f ms'%((t1-t)*1e3,(t2-t1)*1e3))
