"""Where a V-cycle of the demos' sizes goes, leg by leg (HIP events, gmg_profile_cycle): levels, host coarsest round trip, residual check."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
for n1 in [int(a) for a in sys.argv[1:]] or [190, 390]:
    V, F = meshgen.torus_mesh(n1, n1)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
    lhs, rhs = meshgen.smoothing_system(S, mass, V)
    for kw in ({}, {"coarse_mode": cabi.COARSE_DEVICE_INVERSE}):
        eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        for d, b in ((3, rhs), (1, np.asfortranarray(rhs[:, :1]))):
            eng.load_problem(b, b); eng.run_cycles(5, 2)
            c0 = eng.timing("coarse_host_ms")
            import time; t = time.perf_counter(); eng.run_cycles(50, 2); ms = 1e3 * (time.perf_counter() - t) / 50
            host_us = 1e3 * (eng.timing("coarse_host_ms") - c0) / 50
            legs = eng.profile_cycle(2, 10)
            print(json.dumps({"n": n1 * n1, "d": d, "kw": {k: int(v) for k, v in kw.items()}, "levels": [eng.level_info(k)["n"] for k in range(eng.num_levels + 1)], "ms_per_cycle": round(ms, 4),
                              "host_solve_us": round(host_us, 1), "legs_us": [round(1e3 * float(v), 1) for v in legs]}), flush=True)
        eng.close()
