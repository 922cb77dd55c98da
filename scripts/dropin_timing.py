"""Wall time of the reference's own API (`gravomg.MultigridSolver`, drop-in package) on the bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gravo_mg_amd", "dropin"))
import numpy as np, scipy.sparse as sp
from gravo_mg_amd import meshgen
from gravomg import MultigridSolver
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 1732
V, F = meshgen.torus_mesh(n1, n1)
S, mass = meshgen.cotan_laplacian(V, F)
neigh = meshgen.neighbors_from_stiffness(S)
lhs, rhs = meshgen.poisson_system(S, mass)
lhs = sp.csr_matrix(lhs)
t = time.perf_counter(); s = MultigridSolver(V, neigh, sp.diags(mass).tocsr()); print(f"constructor (hierarchy) {time.perf_counter() - t:.2f} s")
for rep in range(3):
    t = time.perf_counter(); x = s.solve(lhs, rhs); dt = time.perf_counter() - t
    print(f"solve() call {1e3 * dt:.1f} ms; residue {s.residual(lhs, rhs, x) if hasattr(s, 'residual') else float('nan')}")
r = lhs @ x - rhs
print("M-norm residue", float(np.sqrt(((r[:, 0] ** 2) * mass).sum() / ((rhs[:, 0] ** 2) * mass).sum())))
