"""BASELINE config 1 (36 k smoothing, d = 3): steady-state cycles for a rocprofv3 kernel trace."""
import sys, time
sys.path.insert(0, '.')
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(190, 190)
S, mass = meshgen.cotan_laplacian(V, F)
lhs, rhs = meshgen.smoothing_system(S, mass, V)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
eng = cabi.Engine()
eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
print([eng.level_info(k) for k in range(eng.num_levels + 1)])
eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
t = time.perf_counter(); eng.run_cycles(20, 2); print("ms/cycle", 50 * (time.perf_counter() - t), "coarse_host_ms", eng.timing("coarse_host_ms") / 23)
for d in (1, 3):
    eng.load_problem(rhs[:, :d].copy(), rhs[:, :d].copy()); eng.run_cycles(3, 2)
    t = time.perf_counter(); eng.run_cycles(20, 2); print("d", d, "ms/cycle", 50 * (time.perf_counter() - t))
