"""Small meshes (the demos' sizes): level 0 colour-major (default) against blocked (block_from_level = 0) -- cycles to 1e-4, ms per cycle, solve call; smoothing (n x 3) and Poisson (d = 1)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
for n1 in [int(a) for a in sys.argv[1:]] or [190, 390]:
    V, F = meshgen.torus_mesh(n1, n1)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
    for sysname, (lhs, rhs) in (("smoothing d=3", meshgen.smoothing_system(S, mass, V)), ("poisson d=1", meshgen.poisson_system(S, mass))):
        C = lhs.tocoo(); off = C.row != C.col
        for kw in ({}, {"block_from_level": 0}):
            eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
            x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
            ts = []
            for _ in range(5):
                t = time.perf_counter(); eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100); ts.append(1e3 * (time.perf_counter() - t))
            eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
            t = time.perf_counter(); eng.run_cycles(50, 2); ms = 1e3 * (time.perf_counter() - t) / 50
            print(json.dumps({"n": n1 * n1, "system": sysname, "kw": kw, "iters": int(it), "solve_ms_median": round(float(np.median(ts)), 3), "ms_per_cycle": round(ms, 4),
                              "max_offdiag": float(C.data[off].max()), "residues": [float(v) for v in conv[:, 1]][:8]}), flush=True)
            eng.close()
