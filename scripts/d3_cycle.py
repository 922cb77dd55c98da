"""3 M-vertex smoothing system with d = 1, 2, 3, 4 right-hand sides: cycle time and fine-level kernel rates."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(1732, 1732)
S, mass = meshgen.cotan_laplacian(V, F)
lhs, rhs3 = meshgen.smoothing_system(S, mass, V)
rhs = np.column_stack([rhs3, rhs3[:, :1]])
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
kw = dict(a.split('=') for a in sys.argv[1:]); kw = {k: int(v) for k, v in kw.items()}
print(kw)
eng = cabi.Engine(**kw)
eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
for d in (1, 3):
    b = np.ascontiguousarray(rhs[:, :d])
    eng.load_problem(b, b); eng.run_cycles(3, 2)
    t = time.perf_counter(); eng.run_cycles(20, 2); ms = 50 * (time.perf_counter() - t)
    line = f"d={d}: {ms:.3f} ms/cycle"
    for name, kind in (("gs_sweep", 0), ("residual", 1), ("restrict", 2), ("prolong", 3), ("norm", 4)):
        t_ms, launches = eng.bench_kernel(kind, 0, d, 30)
        by = eng.algorithmic_bytes(kind, 0, d)
        line += f" | {name} {1e3 * t_ms:.1f} us {by / t_ms / 1e6:.0f} GB/s"
    for k in (1, 2):
        t_ms, _ = eng.bench_kernel(0, k, d, 30)
        line += f" | L{k} sweep {1e3 * t_ms:.1f} us"
    print(line)
