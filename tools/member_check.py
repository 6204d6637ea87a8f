import sys, numpy as np, scipy.sparse as sp, torch, time
sys.path.insert(0,'spacetime-fullgrid-parallel_amd'); sys.path.insert(0,'.')
from source.assembly import space_matrices
from source.problem import problem_helper
from source.multigrid import MeshHierarchy, MultiGrid, MultiGridFamily
for problem, Js in (('square',4),('square',6),('lshape',5)):
    mesh=problem_helper(problem,J_space=Js,J_time=2)[0]
    M_x,A_x=space_matrices(mesh)
    hier=MeshHierarchy(mesh)
    ca=0.3; cms=[2.0**j for j in range(5)]
    b=np.random.RandomState(0).rand(M_x.shape[0],6)
    for exact in (False,True):
        for kw in (dict(fuse_restrict=False, gs_rows='full'), dict()):
            fam=MultiGridFamily(A_x,M_x,hier,ca=ca,cms=cms,smoothsteps=3,vcycles=2,exact_coarse=exact,**kw)
            devs=[]
            for k in (0,2,4):
                ref=MultiGrid(sp.csr_matrix(cms[k]*M_x+ca*A_x),hier,smoothsteps=3,vcycles=2,**kw)
                y=fam.members[k]@b; yr=ref@b
                devs.append(np.abs(y-yr).max()/np.abs(yr).max())
            print(problem,Js,'exact_coarse',exact,kw,'member levels',fam._dev.member_levels,'max rel dev vs MultiGrid(assembled): %s'%(' '.join('%.1e'%d for d in devs)),flush=True)
import heateq_mpi as hm
for rep in range(2):
    t=time.time(); h=hm.HeatEquationMPI(J_space=9,J_time=6); torch.cuda.synchronize(); print('setup %.2f s'%(time.time()-t)); h.P @ h.rhs; print('member levels', h.C_family._dev.member_levels)
    for label, at in h.setup_timeline: print('   %-48s at %.2f'%(label,at))
    del h
