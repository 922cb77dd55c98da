"""The drop-in surface: `import gravomg` / `gravomg_bindings` from gravo_mg_amd/dropin/ with the reference's names,
signatures and defaults (gravomg_bindings/src/gravomg/core.py:7-147, gravomg_bindings/src/cpp/core.cpp:142-180).
CPU part: import, signature, helpers vs captured reference outputs, loud failure without a GPU.
GPU part (-m gpu): the reference's own call pattern (demos/smoothing.py:35-52, experiments/python/comparisons.py:167-174)."""
import inspect
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "gravo_mg_amd", "dropin")


@pytest.fixture(scope="module")
def gravomg(cabi):
    import glob
    if not glob.glob(os.path.join(DROPIN, "gravomg_bindings*.so")):
        import __graft_entry__
        __graft_entry__.build()
    if DROPIN not in sys.path:
        sys.path.insert(0, DROPIN)
    import gravomg as g
    return g


def _problem():
    from gravo_mg_amd import meshgen
    V, F = meshgen.torus_mesh(48, 40)
    S, mass = meshgen.cotan_laplacian(V, F)
    return V, F, S, sp.diags(mass).tocsr(), mass


def test_api_surface_matches_reference(gravomg):
    sig = inspect.signature(gravomg.MultigridSolver.__init__)
    want = [("pos", inspect._empty), ("neigh", inspect._empty), ("mass", inspect._empty), ("ratio", 8.0), ("lower_bound", 1000),
            ("cycle_type", 0), ("tolerance", 1e-4), ("stopping_criteria", 2), ("pre_iters", 2), ("post_iters", 2), ("max_iter", 100),
            ("check_voronoi", True), ("nested", False), ("sampling_strategy", gravomg.Sampling.FASTDISK),
            ("weighting", gravomg.Weighting.BARYCENTRIC), ("sig06", False), ("normals", None), ("verbose", False), ("debug", False),
            ("ablation", False), ("ablation_num_points", 3), ("ablation_random", False)]
    got = [(n, p.default) for n, p in list(sig.parameters.items())[1:]]
    assert got == want
    for name in ("solve", "direct_solve", "residual", "set_prolongation_matrices", "prolongation_matrices", "write_solver_timing",
                 "write_hierarchy_timing", "write_convergence", "construct_sig21_hierarchy", "toggle_hierarchy", "sampling_indices",
                 "level_points", "level_edges", "notrimap", "all_triangles", "coarse_normals", "nearest_source"):
        assert hasattr(gravomg.MultigridSolver, name), name
    assert [e.name for e in gravomg.Sampling.__members__.values()] == ["FASTDISK", "POISSONDISK", "FPS", "RANDOM", "MIS"]
    assert [e.name for e in gravomg.Weighting.__members__.values()] == ["BARYCENTRIC", "UNIFORM", "INVDIST"]
    assert [e.name for e in gravomg.Hierarchy.__members__.values()] == ["OURS", "SIG21"]
    for fn in ("neighbors_from_stiffness", "neighbors_from_faces", "knn_undirected", "normalize_area", "normalize_bounding_box"):
        assert callable(getattr(gravomg, fn))


def test_util_matches_captured_reference_outputs(gravomg):
    z = np.load(os.path.join(ROOT, "tests", "golden", "util_neigh.npz"))
    n = z["V"].shape[0]
    S = sp.csc_matrix((z["S_data"], z["S_indices"], z["S_indptr"]), shape=(n, n))
    for fmt in ("csc", "csr"):
        assert np.array_equal(gravomg.neighbors_from_stiffness(S.asformat(fmt)), z["neigh_from_stiffness"])
    assert np.array_equal(gravomg.neighbors_from_faces(z["F"]), z["neigh_from_faces"])
    np.testing.assert_allclose(gravomg.normalize_area(z["V"] * 3.7 + 1.0, z["F"]), z["normalize_area"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(gravomg.normalize_bounding_box(z["P"]), z["normalize_bounding_box"], rtol=1e-13, atol=1e-15)
    kn = gravomg.knn_undirected(z["P"], 3)
    assert kn.dtype == np.int32 and kn.shape[0] == z["P"].shape[0]
    for i, row in enumerate(kn):                                   # symmetric, self excluded
        for j in row[row >= 0]:
            assert i != j and i in kn[j]


def test_construction_and_hierarchy_without_gpu(gravomg, tmp_path):
    """Hierarchy construction is host-side: works on a CPU box; the solve itself must refuse loudly there."""
    V, F, S, M, mass = _problem()
    neigh = gravomg.neighbors_from_stiffness(S)
    solver = gravomg.MultigridSolver(V, neigh, M, lower_bound=40)
    U = solver.prolongation_matrices
    assert len(U) >= 2 and all(sp.issparse(u) for u in U) and U[0].shape[0] == V.shape[0]
    f = tmp_path / "hier.csv"
    solver.write_hierarchy_timing("torus", str(f), True)
    head, row = f.read_text().strip().split("\n")
    assert head.split(",")[0] == "experiment" and {"hierarchy", "n_vertices", "levels", "sampling", "cluster"} <= set(head.split(","))
    assert row.split(",")[0] == "torus" and len(row.split(",")) == len(head.split(","))
    # what every build keeps beside U (core.cpp:90-92, 114-116) and the debug-only coarse positions (:94-96)
    dof = [V.shape[0]] + [u.shape[1] for u in U]
    samples, nearest = solver.sampling_indices, solver.nearest_source
    assert len(samples) == len(nearest) == len(U) and solver.level_points == []
    for k in range(len(U)):
        assert len(samples[k]) == dof[k + 1] and len(set(samples[k])) == dof[k + 1] and 0 <= min(samples[k]) and max(samples[k]) < dof[k]
        assert len(nearest[k]) == dof[k] and 0 <= min(nearest[k]) and max(nearest[k]) < dof[k + 1]
        assert all(nearest[k][samples[k][c]] == c for c in range(0, dof[k + 1], 7))       # a sample belongs to its own cluster
    dbg = gravomg.MultigridSolver(V, neigh, M, lower_bound=40, debug=True)
    pts = dbg.level_points
    assert [p.shape for p in pts] == [(d, 3) for d in dof[1:]]
    assert np.abs(pts[0]).max() <= np.abs(V).max() * 1.0001                                # cluster centroids stay inside the hull
    # the reference's remaining debug members: levelE (SIG06 hierarchy only) and levelN (never filled) are empty upstream too;
    # allTriangles / noTriFoundMap exist with debug=True only (multigrid_solver.cpp:281, 291)
    for s_ in (solver, dbg):
        assert list(s_.level_edges) == [] and list(s_.coarse_normals) == []
    assert list(solver.all_triangles) == [] and list(solver.notrimap) == []
    tris, ntm = dbg.all_triangles, dbg.notrimap
    assert len(tris) == len(ntm) == len(U)
    for k in range(len(U)):
        t = np.asarray(tris[k])
        assert t.ndim == 2 and t.shape[1] == 3 and t.shape[0] >= dof[k + 1] // 2 and t.min() >= 0 and t.max() < dof[k + 1]
        assert np.all(t[:, 0] != t[:, 1]) and np.all(t[:, 1] != t[:, 2]) and np.all(t[:, 0] != t[:, 2])
        assert len(ntm[k]) == dof[k] and not np.any(ntm[k])
    with pytest.raises(RuntimeError, match="Pardiso"):
        solver.direct_solve((M + 1e-3 * S).tocsr(), M @ V, pardiso=True)
    with pytest.raises(TypeError):
        solver.write_convergence()
    with pytest.raises(RuntimeError):
        gravomg.MultigridSolver(V, neigh, M, lower_bound=40, sampling_strategy=gravomg.Sampling.MIS)
    with pytest.raises(RuntimeError):
        gravomg.MultigridSolver(V, neigh, M, lower_bound=40, sig06=True)
    from gravo_mg_amd import cabi
    if cabi.device_count() == 0:
        lhs = (M + 1e-3 * S).tocsr()
        with pytest.raises(RuntimeError, match="no usable HIP device"):
            solver.solve(lhs, M @ V)


@pytest.mark.gpu
def test_reference_call_pattern_smoothing_and_poisson(gravomg, oracle, tmp_path):
    V, F, S, M, mass = _problem()
    neigh = gravomg.neighbors_from_stiffness(S)
    solver = gravomg.MultigridSolver(V, neigh, M, verbose=False, ratio=8, lower_bound=40, tolerance=1e-4, max_iter=100,
                                     sampling_strategy=gravomg.Sampling.FASTDISK)
    # demos/smoothing.py:43-52
    lhs = (M + 0.001 * S).tocsr()
    rhs = M @ V
    x = solver.solve(lhs, rhs)
    assert x.shape == (V.shape[0], 3) and x.flags.f_contiguous
    res = solver.residual(lhs, rhs, x)
    assert res <= 1e-4
    assert abs(res - oracle.residual_check(lhs, mass, rhs, x, 2)) <= 1e-3 * res + 1e-9
    t = solver.solver_timing
    assert {"reduction", "coarsest_solve", "cycles", "solver_total", "iterations", "residue"} <= set(t)
    assert t["iterations"] >= 1 and t["residue"] <= 1e-4 and t["solver_total"] >= t["cycles"] > 0
    # same answer as the CPU restatement of the reference given the same hierarchy
    O = oracle.Hierarchy(solver.prolongation_matrices, mass)
    O.set_system(lhs)
    xo, ito, reso, _ = O.solve(rhs, tol=1e-4)
    assert abs(t["iterations"] - ito) <= 2
    assert np.sqrt((mass[:, None] * (x - xo) ** 2).sum() / (mass[:, None] * xo ** 2).sum()) <= 2e-3
    # experiments/python/comparisons.py:75-96,167-174: Poisson, rhs = M y, CSV writers
    lhs_p = (M * 1e-6 + S).tocsr()
    y = np.random.default_rng(42).standard_normal((V.shape[0], 1))
    xp = solver.solve(lhs_p, M @ y)
    assert solver.residual(lhs_p, M @ y, xp) <= 1e-4
    f1, f2 = tmp_path / "solver.csv", tmp_path / "conv.csv"
    solver.write_solver_timing("torus", str(f1), True)
    solver.write_solver_timing("torus", str(f1), False)
    lines = f1.read_text().strip().split("\n")
    assert lines[0].startswith("experiment,") and len(lines) == 3 and "iterations" in lines[0] and "solver_total" in lines[0]
    solver.write_convergence(str(f2))
    conv = f2.read_text().strip().split("\n")
    assert conv[0] == "time,residue" and len(conv) - 1 == len(solver.convergence)
    # injected hierarchy + other norms + unsupported options
    solver.set_prolongation_matrices(solver.prolongation_matrices[:1])
    x1 = solver.solve(lhs, rhs)
    assert solver.residual(lhs, rhs, x1, 0) <= 1e-3
    with pytest.raises(RuntimeError):
        gravomg.MultigridSolver(V, neigh, M, lower_bound=40, cycle_type=1).solve(lhs, rhs)
    xd = solver.direct_solve(lhs, rhs)
    assert np.linalg.norm(lhs @ xd - rhs) <= 1e-10 * np.linalg.norm(rhs)


_FALLBACK_SCRIPT = r"""
import io, os, sys, contextlib
import numpy as np, scipy.sparse as sp
root = sys.argv[1]
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "gravo_mg_amd", "dropin"))
sys.path.insert(0, os.path.join(root, "tests", "_native"))        # gravomg_bindings built with -DGMG_TESTING shadows the production module
import gravomg_bindings
assert "tests/_native" in gravomg_bindings.__file__.replace(os.sep, "/"), gravomg_bindings.__file__
import gravomg
from gravo_mg_amd import meshgen
V, F = meshgen.torus_mesh(48, 40)
S, mass = meshgen.cotan_laplacian(V, F)
M = sp.diags(mass).tocsr()
solver = gravomg.MultigridSolver(V, gravomg.neighbors_from_stiffness(S), M, lower_bound=40, tolerance=1e-4, max_iter=30)
lhs = (M * 1e-6 + S).tocsr()
rhs = M @ np.random.default_rng(1).standard_normal((V.shape[0], 1))
def rel(a, b): return float(np.sqrt((mass[:, None] * (a - b) ** 2).sum() / (mass[:, None] * b ** 2).sum()))
x_ok = solver.solve(lhs, rhs)
assert solver.solver_timing["fallback_exact_gs"] == 0.0 and solver.residual(lhs, rhs, x_ok) <= 1e-4
solver.solver._test_report_diverged(1)                              # the next default-engine solve reports GMG_DIVERGED
x = solver.solve(lhs, rhs)
t = solver.solver_timing
assert t["fallback_exact_gs"] == 1.0 and t["diverged"] == 0.0, t
assert t["residue"] <= 1e-4 and solver.residual(lhs, rhs, x) <= 1e-4 and rel(x, x_ok) <= 1e-2
# the safe configuration stays for THIS system (no second failed attempt) ...
solver.solve(lhs, rhs)
assert solver.solver_timing["fallback_exact_gs"] == 1.0 and solver.solver_timing["residue"] <= 1e-4
# ... and only for it: another matrix runs the configured engine again
lhs2 = (M * 1e-3 + S).tocsr()
x2 = solver.solve(lhs2, rhs)
assert solver.solver_timing["fallback_exact_gs"] == 0.0 and solver.residual(lhs2, rhs, x2) <= 1e-4
# back to the first system: it is remembered as one that needs the fallback (no second failed attempt), and the engine that solved it was
# parked, not destroyed
x_again = solver.solve(lhs, rhs)
assert solver.solver_timing["fallback_exact_gs"] == 1.0 and np.array_equal(x_again, x)
x2_again = solver.solve(lhs2, rhs)
assert solver.solver_timing["fallback_exact_gs"] == 0.0 and np.array_equal(x2_again, x2)
# the verdict was reached under the options of that time: after the caller changes them, the remembered system gets the CONFIGURED engine again
solver.set_engine_option("gs_omega", 1.2)
x3 = solver.solve(lhs, rhs)
assert solver.solver_timing["fallback_exact_gs"] == 0.0 and solver.residual(lhs, rhs, x3) <= 1e-4
solver.set_engine_option("gs_omega", 1.35)
assert solver.solve(lhs, rhs) is not None and solver.solver_timing["fallback_exact_gs"] == 1.0        # (the old options: remembered)
# a smoother the caller chose is never replaced -- not by a fresh failure, and not through the memory of the remembered system either:
# weighted Jacobi far beyond its stability limit blows up on BOTH matrices, and solve() says so
solver.set_engine_option("smoother", 1)
solver.set_engine_option("jacobi_omega", 1.95)
for m in (lhs2, lhs):
    try:
        solver.solve(m, rhs)
        raise SystemExit("expected the Jacobi iteration to be reported as diverged")
    except RuntimeError as e:
        assert "diverged" in str(e), e
    assert solver.solver_timing["fallback_exact_gs"] == 0.0
print("FALLBACK-OK")
"""


@pytest.mark.gpu
def test_diverging_smoother_falls_back_to_gauss_seidel_on_every_level(gravomg):
    """The engine's default smoothers carry no convergence guarantee for every SPD matrix; solve() must notice an iteration that
    does not contract and repeat it, from the same initial guess, with Gauss-Seidel (colour order) on every level -- for THAT
    system and THOSE engine options only, with a message, without touching the caller's engine options, and never in place of a
    smoother the caller chose.  No SPD system here makes the default smoothers fail (a CPU model of the cycle on shifted adjacency
    and signless-Laplacian matrices contracts as well), so the first failure is injected: a build of the pybind module with
    -DGMG_TESTING (tests/_native/, made by csrc/build_bindings.sh; the production module and libgravomg_hip.so hold no hook) lets
    the next default-engine solve report GMG_DIVERGED.  Runs in a subprocess: this process already holds the production module."""
    import subprocess
    out = subprocess.run([sys.executable, "-c", _FALLBACK_SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "FALLBACK-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "Gauss-Seidel in colour order on every level" in out.stdout        # said so without `verbose`
    assert out.stdout.count("did not contract on this system") == 1           # one failed attempt in all: the verdict is remembered



@pytest.mark.gpu
def test_system_matrix_storage_formats_give_the_same_answers(gravomg, oracle):
    """The shim maps CSR storage in place, CSC storage of a symmetric matrix too (verified on every entry), and converts
    everything else (COO, ...): same results on every route."""
    V, F, S, M, mass = _problem()
    neigh = gravomg.util.neighbors_from_stiffness(S)
    solver = gravomg.MultigridSolver(V, neigh, M, lower_bound=60)
    lhs = sp.csr_matrix(M + 1e-3 * S)
    rhs = M @ V
    x_csr = solver.solve(lhs, rhs)
    x_csc = solver.solve(sp.csc_matrix(lhs), rhs)          # mapped in place
    x_coo = solver.solver.solve(sp.coo_matrix(lhs), rhs)    # straight into the pybind11 shim
    assert np.array_equal(x_csr, x_csc) and np.array_equal(x_csr, x_coo)
    assert abs(solver.residual(lhs, rhs, x_csr) - oracle.residual_check(lhs, mass, rhs, x_csr, 2)) <= 1e-9
    assert solver.residual(lhs, rhs, x_csr) == solver.residual(sp.csc_matrix(lhs), rhs, x_csr)


@pytest.mark.gpu
def test_slightly_unsymmetric_csr_is_not_taken_for_its_transpose(gravomg):
    """The engine works on the outer vectors of the storage it is handed as rows.  CSR storage is therefore used in place for
    any matrix; CSC storage only when EVERY entry has a symmetric partner (full check, cached by content digest).  Three
    overwritten entries (the Dirichlet-row pattern) must send a CSC matrix through the real conversion: the residual the
    shim reports is that of A on every route, not of A^T."""
    V, F, S, M, mass = _problem()
    neigh = gravomg.util.neighbors_from_stiffness(S)
    solver = gravomg.MultigridSolver(V, neigh, M, lower_bound=60)
    lhs = sp.csr_matrix(M + 1e-3 * S)
    bad = lhs.copy()
    for r in (5, 700, 1500):                       # scale one off-diagonal entry of three rows, not its mirror image
        p = bad.indptr[r] + (1 if bad.indices[bad.indptr[r]] == r else 0)
        bad.data[p] *= 1.5
    rhs = M @ V
    x = np.random.default_rng(3).standard_normal(rhs.shape)
    for t, want in ((0, None), (3, np.linalg.norm(bad @ x - rhs))):
        got = solver.residual(bad, rhs, x, t)
        assert got == solver.residual(sp.csc_matrix(bad), rhs, x, t)             # same route as an explicit conversion
        if want is not None:
            assert abs(got - want) <= 1e-12 * want
            assert abs(got - np.linalg.norm(bad.T @ x - rhs)) > 1e-6 * want       # ... and measurably not the transpose's
    assert solver.residual(lhs, rhs, x, 3) == solver.residual(sp.csc_matrix(lhs), rhs, x, 3)
