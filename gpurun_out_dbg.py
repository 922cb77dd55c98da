import sys; sys.path.insert(0,'.')
import numpy as np, scipy.sparse as sp
from tests import problems
from gravo_mg_amd import cabi
from oracle import oracle
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
for name,P in [("poisson",problems.torus_problem(48,40,"poisson",60)),("smooth",problems.torus_problem(40,36,"smoothing",60)),("pc",problems.pointcloud_problem(3000))]:
    eng=cabi.Engine(); eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    L=len(P.U); orders=[]
    for k in range(L):
        n2o,_=eng.level_ordering(k); orders.append(n2o[n2o>=0])
    orders.append(np.arange(P.U[-1].shape[1]))
    Up=[sp.csc_matrix(P.U[k].tocsr()[orders[k]][:,orders[k+1]]) for k in range(L)]
    lhs_p=sp.csc_matrix(P.lhs.tocsr()[orders[0]][:,orders[0]])
    O=oracle.Hierarchy(Up,P.mass[orders[0]]); O.set_system(lhs_p)
    got=eng.vcycle(P.rhs,P.rhs.copy()); wp=O.vcycle(P.rhs[orders[0]],P.rhs[orders[0]].copy()); want=np.empty_like(wp); want[orders[0]]=wp
    print(name,"L",L,[eng.level_info(k) for k in range(L+1)])
    print(" vcycle rel x",rel(got,want)," rel resid diff", np.linalg.norm(P.lhs@(got-want))/np.linalg.norm(P.rhs))
    for tol in (1e-4,1e-6,1e-8,1e-10):
        x,it,res,conv=eng.solve(P.rhs,tol=tol,max_iter=60)
        O2=oracle.Hierarchy(P.U,P.mass); O2.set_system(P.lhs); xo,ito,reso,_=O2.solve(P.rhs,tol=tol,max_iter=60)
        m=P.mass[:,None]
        print("  tol",tol,"gpu",it,res,"cpu",ito,reso,"dxM",np.sqrt((m*(x-xo)**2).sum()/(m*xo**2).sum()), "xnorm", np.linalg.norm(xo)/np.linalg.norm(P.rhs))
