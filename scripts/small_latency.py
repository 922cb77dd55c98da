"""Fixed costs on a small problem (BASELINE config 1, 36 k vertices, d = 3): set_system and solve per fresh engine."""
import sys, time
sys.path.insert(0, '.')
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(190, 190)
S, mass = meshgen.cotan_laplacian(V, F)
lhs, rhs = meshgen.smoothing_system(S, mass, V)
t = time.perf_counter(); H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S)); print(f"hierarchy {1e3 * (time.perf_counter() - t):.1f} ms")
for rep in range(3):
    t = time.perf_counter(); eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); t1 = time.perf_counter()
    eng.set_system(lhs); t2 = time.perf_counter()
    x, it, res, conv = eng.solve(rhs); t3 = time.perf_counter()
    eng.set_system(lhs); t4 = time.perf_counter()
    x, it, res, conv = eng.solve(rhs); t5 = time.perf_counter()
    keys = ["pattern_key", "upload_A0", "rap_l1", "ordering_ready_l0", "ordering_ready_l1", "device_layout", "factor_joined", "mass_done"]
    tl = []
    for k in keys:
        try: tl.append((k, round(eng.timing("t_" + k), 2)))
        except Exception: pass
    print(f"engine {rep}: create+hierarchy {1e3 * (t1 - t):.1f} ms, set_system {1e3 * (t2 - t1):.1f} ms, solve {1e3 * (t3 - t2):.1f} ms ({it} it), repeat set_system {1e3 * (t4 - t3):.1f} ms, solve {1e3 * (t5 - t4):.1f} ms; factor {eng.timing('coarsest_solve'):.1f} ms")
    print("   repeat timeline", tl)
    del eng
