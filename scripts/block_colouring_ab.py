"""In-block colouring order of the block sweeps: breadth-first (default) against smallest-last (GMG_BLOCK_COLOURING=sl), one process per setting.
  python scripts/block_colouring_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gravo_mg_amd import cabi, meshgen
tag = os.environ.get("GMG_BLOCK_COLOURING", "sl")

def run(name, H, mass, lhs, rhs, steps=100, tol=1e-4):
    eng = cabi.Engine()
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    eng.load_problem(rhs, rhs); eng.run_cycles(10, 2)
    t = time.perf_counter(); eng.run_cycles(steps, 2); cyc = 1e3 * (time.perf_counter() - t) / steps
    x, it, res, conv = eng.solve(rhs, tol=tol, max_iter=200)
    cols = [eng.level_info(k).get("n_colors") for k in range(eng.num_levels + 1)]
    legs = eng.profile_cycle(2, 10)
    print(f"{tag} {name}: {cyc:.4f} ms per cycle, {it} cycles to {tol:g} (residue {res:.3e}), max colours per level {cols}, legs {[round(float(v), 4) for v in legs]}", flush=True)
    eng.close()

V, F = meshgen.torus_mesh(1732, 1732)
S, mass = meshgen.cotan_laplacian(V, F)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=1000)
lhs, rhs = meshgen.poisson_system(S, mass)
run("3M d1", H, mass, lhs, rhs)
lhs3, rhs3 = meshgen.smoothing_system(S, mass, V)
run("3M d3", H, mass, lhs3, rhs3)
P = meshgen.torus_points(2000000, noise=0.002)
Sp, mp = meshgen.knn_graph_laplacian(P, 8)
Hp = cabi.Hierarchy(P, meshgen.neighbors_from_stiffness(Sp), lower_bound=1000)
lp, rp = meshgen.poisson_system(Sp, mp)
run("2M point cloud", Hp, mp, lp, rp)
