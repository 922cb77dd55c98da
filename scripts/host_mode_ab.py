"""Host coarsest solve (GMG_COARSE_HOST_LDLT, the drop-in class's default) with the head of the next cycle on / off and the stream gate on / off.
  python scripts/host_mode_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gravo_mg_amd import cabi, meshgen

def run(tag, n1, n2, kind, d, steps=50, reps=4):
    V, F = meshgen.torus_mesh(n1, n2)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=1000)
    lhs, rhs = (meshgen.smoothing_system(S, mass, V) if kind == "smoothing" else meshgen.poisson_system(S, mass, d=d))
    for spec, gate in ((1, 1), (0, 1), (1, 0), (0, 0)):
        eng = cabi.Engine(coarse_mode=cabi.COARSE_HOST_LDLT, speculate_head=spec, stream_gate=gate)
        eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        eng.load_problem(rhs, rhs); eng.run_cycles(10, 2)
        ms = []
        for r in range(reps):
            eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
            t = time.perf_counter(); eng.run_cycles(steps, 2); ms.append(1e3 * (time.perf_counter() - t) / steps)
        out = np.empty(np.asfortranarray(rhs).shape, order="F")
        sol = []
        for r in range(reps):
            t = time.perf_counter(); x, it, res, conv = eng.solve(np.asfortranarray(rhs), tol=1e-4, out=out); sol.append((round(1e3 * (time.perf_counter() - t), 3), it, round(eng.timing("cycles"), 3)))
        legs = eng.profile_cycle(2, 5)
        print(f"{tag} head={spec} gate={gate}: ms per cycle {[round(v, 4) for v in ms]}; solve {sol}; legs sum {float(np.sum(legs)):.4f}", flush=True)
        eng.close()

if __name__ == "__main__":
    run("36k d3", 190, 190, "smoothing", 3)
    run("722k d1", 850, 850, "poisson", 1)
    run("3M d1", 1732, 1732, "poisson", 1)
