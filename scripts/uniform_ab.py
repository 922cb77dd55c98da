"""Equally wide slices read without their pointers (gmg_config::uniform_slices) on and off, same process, alternating timed loops.
  python scripts/uniform_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gravo_mg_amd import cabi, meshgen

def run(tag, n1, n2, kind, d, steps=200, reps=5):
    V, F = meshgen.torus_mesh(n1, n2)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=1000)
    lhs, rhs = (meshgen.smoothing_system(S, mass, V) if kind == "smoothing" else meshgen.poisson_system(S, mass, d=d))
    engs = {}
    for u in (1, 0):
        eng = cabi.Engine(uniform_slices=u)
        eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        eng.load_problem(rhs, rhs); eng.run_cycles(20, 2)
        engs[u] = eng
    print(tag, "uniform widths A / R / P:", [engs[1].timing(k + "_uniform_width") for k in ("col16_l0", "col16_R_l0", "col16_P_l0")], flush=True)
    ms = {1: [], 0: []}
    for r in range(reps):
        for u in (1, 0):
            eng = engs[u]
            eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
            t = time.perf_counter(); eng.run_cycles(steps, 2); ms[u].append(1e3 * (time.perf_counter() - t) / steps)
    for u in (1, 0):
        legs = engs[u].profile_cycle(2, 10)
        print(f"{tag} uniform_slices={u}: ms per cycle {[round(v, 4) for v in ms[u]]} median {np.median(ms[u]):.4f}; legs {[round(float(v), 4) for v in legs]}", flush=True)
    for e in engs.values():
        e.close()

if __name__ == "__main__":
    run("3M d1", 1732, 1732, "poisson", 1)
    run("3M d3", 1732, 1732, "smoothing", 3, steps=100)
    run("722k d1", 850, 850, "poisson", 1)
