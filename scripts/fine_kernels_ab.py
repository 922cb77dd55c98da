"""Fine-level kernels of the 3 M-vertex bench workload (restriction, norm, residual, prolongation) for engine variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(1732, 1732)
S, mass = meshgen.cotan_laplacian(V, F)
lhs, rhs = meshgen.poisson_system(S, mass)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
kw = dict(a.split('=') for a in sys.argv[1:]); kw = {k: int(v) for k, v in kw.items()}
eng = cabi.Engine(**kw)
eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
line = f"{kw} R_LANES={os.environ.get('GMG_R_LANES', '-')}"
for k in (0, 1):
    for name, kind in (("sweep", 0), ("residual", 1), ("restrict", 2), ("prolong", 3)) + ((("norm", 4),) if k == 0 else ()):
        t_ms, launches = eng.bench_kernel(kind, k, 1, 50)
        by = eng.algorithmic_bytes(kind, k, 1)
        line += f" | L{k} {name} {1e3 * t_ms:.1f} us {by / t_ms / 1e6:.0f} GB/s"
eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
t = time.perf_counter(); res = eng.run_cycles(20, 2); line += f" | cycle {50 * (time.perf_counter() - t):.3f} ms res {res[-1]:.3e}"
print(line, flush=True)
