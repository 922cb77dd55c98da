"""Back-substitution time of the coarsest-level solver on the 3 M-vertex bench hierarchy (host only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(1732, 1732); S, mass = meshgen.cotan_laplacian(V, F)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S)); A, rhs = meshgen.poisson_system(S, mass)
for U in H.U:
    A = cabi.host_galerkin(A, U)
b = np.random.default_rng(0).standard_normal(A.shape[0])
x, nnz = cabi.host_ldlt_solve(A, b)
print("n", A.shape[0], "nnz(L)", nnz, "residual", np.linalg.norm(A @ x - b) / np.linalg.norm(b))
print(cabi.host_ldlt_probe(A, b, reps=200))      # gravomg_hip_internal.h: timings on 1 .. 8 threads, bitwise agreement, supernodal vs simplicial
