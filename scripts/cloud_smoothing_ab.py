"""The point-cloud demo's call (demos/conformal_flow_pointcloud.py: M + tau S on a kNN Laplacian, n x 3 right-hand side) with level 0 blocked (default for kNN
operators) against colour-major (block_fine = 0): cycles to 1e-4, ms per cycle, solve call -- and the Poisson system of BASELINE config 3 beside it."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
P = meshgen.torus_points(n, noise=0.0005)
S, mass = meshgen.knn_graph_laplacian(P, 8)
H = cabi.Hierarchy(P, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
for name, (lhs, rhs) in (("smoothing tau=1e-3 d=3", meshgen.smoothing_system(S, mass, P)), ("smoothing tau=1e-1 d=3", meshgen.smoothing_system(S, mass, P, tau=1e-1)), ("poisson d=1", meshgen.poisson_system(S, mass))):
    for kw in ({}, {"block_fine": 0}, {"block_fine": 0, "gs_omega": 1.0}):
        eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass)
        t = time.perf_counter(); eng.set_system(lhs); ts = 1e3 * (time.perf_counter() - t)
        x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
        t = time.perf_counter(); eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100); tv = 1e3 * (time.perf_counter() - t)
        eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
        t = time.perf_counter(); eng.run_cycles(20, 2); ms = 1e3 * (time.perf_counter() - t) / 20
        print(json.dumps({"n": n, "system": name, "kw": kw, "blocked0": eng.level_blocks(0) is not None, "iters": int(it), "solve_ms": round(tv, 2), "ms_per_cycle": round(ms, 4), "set_system_ms": round(ts, 1),
                          "residues": [float("%.3g" % v) for v in conv[:, 1]][:8]}), flush=True)
        eng.close()
