"""The reference's demo sizes (demos/smoothing.py: ~35 k vertices, n x 3 right-hand side, a new tau per frame) through the C-ABI: where the
milliseconds of a frame go -- set_system (cold / same pattern), solve call, its cycles -- for tori of 36 k .. 722 k vertices."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
for n1 in [int(a) for a in sys.argv[1:]] or [190, 390, 850]:
    V, F = meshgen.torus_mesh(n1, n1)
    S, mass = meshgen.cotan_laplacian(V, F)
    t = time.perf_counter(); H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000); th = time.perf_counter() - t
    eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass)
    rec = {"n": n1 * n1, "hierarchy_ms": 1e3 * th, "frames": []}
    for frame, tau in enumerate([1e-3, 2e-3, 5e-4, 1e-3]):
        lhs, rhs = meshgen.smoothing_system(S, mass, V, tau=tau)
        t = time.perf_counter(); eng.set_system(lhs); ts = time.perf_counter() - t
        t = time.perf_counter(); x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100); tv = time.perf_counter() - t
        rec["frames"].append({"set_system_ms": round(1e3 * ts, 3), "values_only": eng.timing("setup_values_only"), "solve_ms": round(1e3 * tv, 3), "iters": int(it),
                              "cycles_ms": round(eng.timing("cycles"), 3), "load_ms": round(eng.timing("solve_load"), 3), "fetch_ms": round(eng.timing("solve_fetch"), 3),
                              "setup_keys": {k: round(eng.timing(k), 2) for k in ("reduction", "coarsest_solve", "upload", "setup_total")}})
    eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
    t = time.perf_counter(); eng.run_cycles(50, 2); rec["ms_per_cycle_d3"] = 1e3 * (time.perf_counter() - t) / 50
    rec["levels"] = [eng.level_info(k)["n"] for k in range(eng.num_levels + 1)]
    print(json.dumps(rec), flush=True)
    eng.close()
