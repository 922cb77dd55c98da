import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench as single
from gravo_mg_amd import cabi
H, mass, lhs, rhs = single.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
for i in range(4):
    x, it, res, conv = eng.solve(rhs)
    print({k: round(eng.timing(k), 3) for k in ("solve_call", "solve_load", "load_vectors", "load_b", "load_x", "load_sync", "cycles", "solve_fetch")})
