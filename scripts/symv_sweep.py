"""Coarsest leg (dense_symv) for rows per wave x strides per trip, one process per setting (the choice is read once): run by scripts/r06_job_symv.sh.
  GMG_SYMV_ROWS=.. GMG_SYMV_STRIDES=.. python scripts/symv_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gravo_mg_amd import cabi, meshgen

def run(tag, n1, n2, kind, d):
    V, F = meshgen.torus_mesh(n1, n2)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=1000)
    lhs, rhs = (meshgen.smoothing_system(S, mass, V) if kind == "smoothing" else meshgen.poisson_system(S, mass, d=d))
    eng = cabi.Engine()
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    eng.load_problem(rhs, rhs); eng.run_cycles(10, 2)
    t = time.perf_counter(); eng.run_cycles(100, 2); cyc = 1e3 * (time.perf_counter() - t) / 100
    legs = eng.profile_cycle(2, 20)
    L = eng.num_levels
    print(f"rows {os.environ.get('GMG_SYMV_ROWS', '-')} strides {os.environ.get('GMG_SYMV_STRIDES', '-')} {tag}: n_L={eng.level_info(L)['n']} coarsest leg {1e3 * legs[L]:.2f} us, cycle {cyc:.4f} ms", flush=True)
    eng.close()

if __name__ == "__main__":
    run("36k d3", 190, 190, "smoothing", 3)
    run("152k d3", 390, 390, "smoothing", 3)
    run("722k d1", 850, 850, "poisson", 1)
    run("3M d1", 1732, 1732, "poisson", 1)
    run("3M d3", 1732, 1732, "smoothing", 3)
    run("36k d1", 190, 190, "poisson", 1)
