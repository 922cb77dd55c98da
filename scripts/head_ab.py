"""The head of the next cycle (gmg_config::speculate_head) on and off, same process, alternating timed loops, plus whole solve calls.
  python scripts/head_ab.py"""
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
    for spec in (1, 0):
        eng = cabi.Engine(speculate_head=spec)
        eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        eng.load_problem(rhs, rhs); eng.run_cycles(20, 2)
        engs[spec] = eng
    ms = {1: [], 0: []}
    for r in range(reps):
        for spec in (1, 0):
            eng = engs[spec]
            eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
            t = time.perf_counter(); eng.run_cycles(steps, 2); ms[spec].append(1e3 * (time.perf_counter() - t) / steps)
    sol = {1: [], 0: []}
    out = np.empty(np.asfortranarray(rhs).shape, order="F")
    for r in range(reps):
        for spec in (1, 0):
            eng = engs[spec]
            t = time.perf_counter(); x, it, res, conv = eng.solve(np.asfortranarray(rhs), tol=1e-4, out=out); sol[spec].append((1e3 * (time.perf_counter() - t), it, eng.timing("cycles")))
    for spec in (1, 0):
        print(f"{tag} speculate_head={spec}: ms per cycle {[round(v, 4) for v in ms[spec]]} median {np.median(ms[spec]):.4f}; solve call ms / iterations / loop ms "
              f"{[(round(a, 3), b, round(c, 3)) for a, b, c in sol[spec]]}", flush=True)
    for e in engs.values():
        e.close()

if __name__ == "__main__":
    run("3M d1", 1732, 1732, "poisson", 1)
    run("3M d3", 1732, 1732, "smoothing", 3, steps=100)
    run("722k d1", 850, 850, "poisson", 1)
    run("36k d3", 190, 190, "smoothing", 3)
