"""Does the block-hybrid coarse smoother ever fail where exact (multicolour) Gauss-Seidel on every level works?  Bilaplacian
smoothing systems M + tau S M^-1 S (not diagonally dominant) over tau, default engine vs block_rows=0, gs_omega=1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 330
V, F = meshgen.torus_mesh(n1, n1)
S, mass = meshgen.cotan_laplacian(V, F)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
B = meshgen.bilaplacian(S, mass)
for tau in (1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
    lhs, rhs = meshgen.smoothing_system(B, mass, V[:, :1], tau=tau)
    line = f"n={V.shape[0]} tau={tau:g}"
    for name, kw in (("default", {}), ("omega=1", dict(gs_omega=1.0)), ("exact GS everywhere", dict(block_rows=0, gs_omega=1.0))):
        eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        eng.load_problem(rhs, rhs)
        r = eng.run_cycles(40, 2)
        line += f" | {name}: r1 {r[0]:.2e} r10 {r[9]:.2e} r40 {r[-1]:.2e}"
    print(line, flush=True)
