#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/*.npz.  Run in the BUILD container only.

Two kinds of fixtures:

1. `vcycle_*.npz` -- known-answer vectors for the hot path.  The reference ships no tests and cannot be built
   here, so the expected outputs come from an INDEPENDENT scipy implementation of the algorithm
   in this file (CSR matvec, `spsolve_triangular` for exact lexicographic Gauss-Seidel, `splu` for the coarsest
   solve) -- deliberately sharing no code with oracle/gravomg_oracle.c or with the HIP engine.  The oracle is
   pinned against these vectors (tests/test_golden.py), and so is the GPU path.
   Inputs stored: A (CSC), U_k (CSC), mass, b.  Expected: x after one pre-smoothing call (2 sweeps), residual,
   restricted residual, coarsest solve, x after prolongation, x after one V-cycle, the residual history for 10
   cycles in all four norm types, and the solution after 10 cycles.

2. `util_neigh.npz` -- captured I/O of the reference's own pure-Python helper
   gravomg_bindings/src/gravomg/util.py (importable here): neighbors_from_stiffness / neighbors_from_faces /
   normalize_area / normalize_bounding_box on a tiny mesh.  Data only -- no reference source is copied.
"""
import importlib.util
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gravo_mg_amd import cabi, meshgen  # noqa: E402  (inputs only: mesh + hierarchy generator)


def gs(A, b, x, iters):
    """Exact forward lexicographic Gauss-Seidel: x <- x + (D+L)^-1 (b - A x), per column."""
    A = sp.csr_matrix(A)
    DL = sp.tril(A, 0).tocsr()
    x = x.copy()
    for _ in range(iters):
        for c in range(x.shape[1]):
            x[:, c] = x[:, c] + spla.spsolve_triangular(DL, b[:, c] - A @ x[:, c], lower=True)
    return x


def norms(A, mass, b, x):
    r = A @ x - b
    out = []
    out.append(max(np.linalg.norm(r[:, c]) / np.linalg.norm(b[:, c]) for c in range(b.shape[1])))
    out.append(max(np.sqrt((r[:, c] ** 2 / mass).sum() / (b[:, c] ** 2 / mass).sum()) for c in range(b.shape[1])))
    out.append(max(np.sqrt((r[:, c] ** 2 * mass).sum() / (b[:, c] ** 2 * mass).sum()) for c in range(b.shape[1])))
    out.append(np.linalg.norm(r))
    return out


def vcycle(As, Us, lu, b, x, k=0, trace=None):
    x = gs(As[k], b, x, 2)
    if trace is not None and k == 0:
        trace["x_pre"] = x.copy()
    r = b - As[k] @ x
    rc = Us[k].T @ r
    if trace is not None and k == 0:
        trace["res"] = r.copy(); trace["rc"] = rc.copy()
    if k == len(Us) - 1:
        e = np.column_stack([lu.solve(rc[:, c]) for c in range(rc.shape[1])])
        if trace is not None:
            trace["coarse_rhs"] = rc.copy(); trace["coarse_sol"] = e.copy()
    else:
        e = vcycle(As, Us, lu, rc, np.zeros_like(rc), k + 1, trace)
    x = x + Us[k] @ e
    if trace is not None and k == 0:
        trace["x_prolonged"] = x.copy()
    return gs(As[k], b, x, 2)


def make_vcycle_fixture(name, pos, S, mass, lhs, rhs, lower_bound):
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(pos, neigh, lower_bound=lower_bound)
    Us = [sp.csc_matrix(u) for u in H.U]
    As = [sp.csr_matrix(lhs)]
    for u in Us:
        As.append(sp.csr_matrix(u.T @ As[-1] @ u))
    lu = spla.splu(sp.csc_matrix(As[-1]))
    b = np.asarray(rhs, dtype=np.float64)
    trace = {}
    x = vcycle(As, Us, lu, b, b.copy(), 0, trace)           # x0 = rhs (core.cpp:69)
    out = {"x_cycle1": x.copy()}
    hist = [norms(As[0], mass, b, x)]
    for _ in range(9):
        x = vcycle(As, Us, lu, b, x)
        hist.append(norms(As[0], mass, b, x))
    out["history"] = np.array(hist)            # [10, 4]: norm types 0..3 after cycles 1..10
    out["x_cycle10"] = x
    out.update(trace)
    A = sp.csc_matrix(lhs); A.sort_indices()
    out.update({"A_indptr": A.indptr.astype(np.int32), "A_indices": A.indices.astype(np.int32), "A_data": A.data, "n": A.shape[0],
                "mass": mass, "b": b, "L": len(Us)})
    for k, u in enumerate(Us):
        u.sort_indices()
        out.update({f"U{k}_indptr": u.indptr.astype(np.int32), f"U{k}_indices": u.indices.astype(np.int32), f"U{k}_data": u.data,
                    f"U{k}_shape": np.array(u.shape)})
        Ak = sp.csc_matrix(As[k + 1]); Ak.sort_indices()
        out.update({f"A{k+1}_indptr": Ak.indptr.astype(np.int32), f"A{k+1}_indices": Ak.indices.astype(np.int32), f"A{k+1}_data": Ak.data})
    np.savez_compressed(os.path.join(HERE, f"vcycle_{name}.npz"), **out)
    print(name, "n", A.shape[0], "levels", [a.shape[0] for a in As], "history[:,2]", out["history"][:, 2])


def make_util_fixture():
    spec = importlib.util.spec_from_file_location("ref_util", "/root/reference/gravomg_bindings/src/gravomg/util.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    V, F = meshgen.torus_mesh(6, 5, jitter=0.3, seed=3)
    S, _ = meshgen.cotan_laplacian(V, F)
    S = sp.csc_matrix(S)                       # the upstream helper is only correct for CSC input
    rng = np.random.default_rng(0)
    P = rng.standard_normal((12, 3))
    np.savez_compressed(os.path.join(HERE, "util_neigh.npz"),
                        V=V, F=F, S_indptr=S.indptr, S_indices=S.indices, S_data=S.data,
                        neigh_from_stiffness=ref.neighbors_from_stiffness(S),
                        neigh_from_faces=ref.neighbors_from_faces(F),
                        normalize_area=ref.normalize_area(V * 3.7 + 1.0, F),
                        P=P, normalize_bounding_box=ref.normalize_bounding_box(P))
    print("util fixture written")


if __name__ == "__main__":
    V, F = meshgen.torus_mesh(24, 20, seed=5)
    S, mass = meshgen.cotan_laplacian(V, F)
    lhs, rhs = meshgen.poisson_system(S, mass, tau=1e-3, seed=1)          # tau=1e-3: well conditioned -> tight vectors
    make_vcycle_fixture("torus480_poisson_d1", V, S, mass, lhs, rhs, 8)            # L = 2
    lhs, rhs = meshgen.smoothing_system(S, mass, V)
    make_vcycle_fixture("torus480_smoothing_d3", V, S, mass, lhs, rhs, 20)
    P = meshgen.torus_points(400, noise=0.002, seed=2)
    S, mass = meshgen.knn_graph_laplacian(P, 6)
    lhs, rhs = meshgen.poisson_system(S, mass, tau=1e-2, seed=9)
    make_vcycle_fixture("pointcloud400_d1", P, S, mass, lhs, rhs, 30)
    make_util_fixture()
