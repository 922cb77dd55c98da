"""Relaxation of a blocked level 0's block sweep (gmg_config::fine_block_omega): V-cycles to 1e-4 / 1e-6 over omega, kNN point clouds, Poisson and smoothing systems."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
omegas = [1.0, 1.1, 1.2, 1.3, 1.4, 1.5]
for n in [int(a) for a in sys.argv[1:]] or [20000, 200000, 2000000]:
    P = meshgen.torus_points(n, noise=0.0005 if n > 100000 else 0.002)
    S, mass = meshgen.knn_graph_laplacian(P, 8)
    H = cabi.Hierarchy(P, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000 if n > 100000 else 200)
    for name, (lhs, rhs) in (("poisson d=1", meshgen.poisson_system(S, mass)), ("smoothing tau=1e-3 d=3", meshgen.smoothing_system(S, mass, P)), ("smoothing tau=1e-1 d=3", meshgen.smoothing_system(S, mass, P, tau=1e-1))):
        row = {"n": n, "system": name, "iters_1e-4": {}, "iters_1e-6": {}, "solve_ms": {}}
        for w in omegas:
            eng = cabi.Engine(fine_block_omega=w); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
            assert eng.level_blocks(0) is not None
            x, it, res, conv = eng.solve(rhs, tol=1e-6, stop_type=2, max_iter=100)
            t = time.perf_counter(); eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100); tv = 1e3 * (time.perf_counter() - t)
            row["iters_1e-6"][w] = int(it) if res <= 1e-6 else -int(it)
            row["iters_1e-4"][w] = int(np.argmax(conv[:, 1] <= 1e-4) + 1) if (conv[:, 1] <= 1e-4).any() else -1
            row["solve_ms"][w] = round(tv, 2)
            eng.close()
        cm = cabi.Engine(block_fine=0); cm.use_hierarchy(H); cm.set_mass(mass); cm.set_system(lhs)
        x, it, res, conv = cm.solve(rhs, tol=1e-6, stop_type=2, max_iter=100)
        row["colour_major"] = {"iters_1e-4": int(np.argmax(conv[:, 1] <= 1e-4) + 1), "iters_1e-6": int(it)}
        cm.close()
        print(json.dumps(row), flush=True)
