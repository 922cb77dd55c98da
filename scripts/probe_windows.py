import sys; sys.path.insert(0,'.')
import numpy as np
from gravo_mg_amd import cabi
from tests import problems
for n,k,lb in ((40000,12,50),(40000,16,50),(60000,20,50),(30000,10,50)):
    P = problems.pointcloud_problem(n,k,lb)
    print(n,k,'levels',[u.shape for u in P.U]);
    eng = cabi.Engine(block_lanes=1); eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    for lev in range(1,len(P.U)):
        e=np.asarray(eng.debug_sell(lev,7)["slice_ptr"]); l=np.asarray(eng.debug_sell(lev,6)["slice_ptr"])
        if len(e)<65: print(n,k,lev,"no ep", eng.level_info(lev)); continue
        print(n,k,lev,eng.level_info(lev)["n"],"E max",(e[64::64]-e[:-64:64]).max(),"L max",(l[64::64]-l[:-64:64]).max(),"nlow max",np.diff(l).max())
    eng.close()
