import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gravo_mg_amd import cabi
from tests import problems
for kind in ("poisson", "smoothing"):
    P = problems.torus_problem(96, 80, "poisson", 30) if kind == "poisson" else problems.torus_problem(64, 60, "smoothing", 60)
    for om in (1.35, 1.0):
        eng = cabi.Engine(use_graph=False, gs_omega=om); eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
        eng.load_problem(P.rhs, P.rhs)
        r = eng.run_cycles(8, 2)
        x = eng.fetch_solution()
        true = eng.residual_norm(P.rhs, x, 2)
        print(kind, om, " ".join(f"{v:.15e}" for v in r), "| api norm of final x", f"{true:.15e}")
