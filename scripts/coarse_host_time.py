"""Steady-state cost of the host coarsest solve inside the V-cycle: ms per cycle and host solve time per cycle over 200 cycles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(1732, 1732); S, mass = meshgen.cotan_laplacian(V, F)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S)); lhs, rhs = meshgen.poisson_system(S, mass)
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
x, it, res, conv = eng.solve(rhs)                      # resets coarse_host_ms
base = eng.timing("coarse_host_ms")
eng.load_problem(rhs, rhs); eng.run_cycles(20, 2)
c0 = eng.timing("coarse_host_ms")
t = time.perf_counter(); eng.run_cycles(200, 2); dt = time.perf_counter() - t
print(f"lib={os.environ.get('GMG_LIB_PATH', 'default')[-24:]}  {1e3 * dt / 200:.4f} ms/cycle; host coarse solve {1e3 * (eng.timing('coarse_host_ms') - c0) / 200:.1f} us per cycle (first solve's 4 cycles: {1e3 * base / it:.1f} us each)")
