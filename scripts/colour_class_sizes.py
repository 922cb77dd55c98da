import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gravo_mg_amd import cabi, meshgen
for cfg in ("4r", "3", "6"):
    name, V, S, mass, lhs, rhs = meshgen.baseline_config(cfg)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
    eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    n2o, cb = eng.level_ordering(0)
    print(name, [int((n2o[cb[c]:cb[c+1]] >= 0).sum()) for c in range(len(cb) - 1)], flush=True)
