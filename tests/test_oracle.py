"""The oracle (oracle/gravomg_oracle.c, the CPU restatement of the reference hot path) against an
independent scipy formulation of the same operators.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from tests import problems


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture(scope="module", params=["poisson", "smoothing-d3", "pointcloud"])
def P(request):
    if request.param == "poisson":
        return problems.torus_problem(48, 40, "poisson", 40)
    if request.param == "smoothing-d3":
        return problems.torus_problem(40, 36, "smoothing", 60)
    return problems.pointcloud_problem(1500, lower_bound=60)


def test_gauss_seidel_is_forward_lexicographic(P, oracle):
    """x <- x + (D+L)^-1 (b - A x): multigrid_solver.cpp:1194-1226."""
    A = sp.csr_matrix(P.lhs)
    DL = sp.tril(A, 0).tocsr()
    rng = np.random.default_rng(0)
    for d in (1, 3):
        b = rng.standard_normal((P.n, d)); x = rng.standard_normal((P.n, d))
        want = x.copy()
        for it in range(1, 4):
            for c in range(d):
                want[:, c] += spla.spsolve_triangular(DL, b[:, c] - A @ want[:, c], lower=True)
            assert rel(oracle.gauss_seidel(P.lhs, b, x, it), want) <= 1e-11
    # 1-D input keeps its shape
    assert oracle.gauss_seidel(P.lhs, b[:, 0], x[:, 0], 1).shape == (P.n,)


def test_gauss_seidel_missing_diagonal_gives_nonfinite(oracle):
    """coeffRef(k,k) on a missing diagonal inserts a zero upstream -> division by zero (multigrid_solver.cpp:1207)."""
    A = sp.csc_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 2.0]]))
    A.eliminate_zeros()
    x = oracle.gauss_seidel(A, np.ones(3), np.zeros(3), 1)
    assert not np.all(np.isfinite(x))


def test_residual_restrict_prolong(P, oracle):
    rng = np.random.default_rng(1)
    U = P.U[0]
    for d in (1, 3):
        b = rng.standard_normal((P.n, d)); x = rng.standard_normal((P.n, d))
        assert rel(oracle.residual(P.lhs, b, x), b - P.lhs @ x) <= 1e-14
        assert rel(oracle.restrict(U, b), U.T @ b) <= 1e-14
        e = rng.standard_normal((U.shape[1], d))
        assert rel(oracle.prolong_add(U, e, x), x + U @ e) <= 1e-14


def test_residual_check_norms(P, oracle):
    """multigrid_solver.cpp:1228-1277."""
    rng = np.random.default_rng(2)
    b = P.rhs; x = rng.standard_normal(b.shape)
    r = P.lhs @ x - b
    m = P.mass[:, None]
    want = [
        max(np.linalg.norm(r[:, c]) / np.linalg.norm(b[:, c]) for c in range(b.shape[1])),
        np.sqrt(((r ** 2 / m).sum(0) / (b ** 2 / m).sum(0)).max()),
        np.sqrt(((r ** 2 * m).sum(0) / (b ** 2 * m).sum(0)).max()),
        np.linalg.norm(r),
    ]
    for t in range(4):
        assert abs(oracle.residual_check(P.lhs, P.mass, b, x, t) - want[t]) <= 1e-12 * want[t]


def test_galerkin_and_coarse_solve(P, oracle):
    """Abar[k] = U^T Abar[k-1] U (multigrid_solver.cpp:1387-1392) and the coarsest direct solve (:1075,1401)."""
    O = oracle.Hierarchy(P.U, P.mass)
    O.set_system(P.lhs)
    A = sp.csc_matrix(P.lhs)
    for k, U in enumerate(P.U):
        A = sp.csc_matrix(U.T @ A @ U)
        Ak = O.level_operator(k + 1)
        assert abs(A - Ak).max() <= 1e-14 * abs(A).max()
    rng = np.random.default_rng(3)
    rc = rng.standard_normal((A.shape[0], 2))
    e = O.coarse_solve(rc)
    assert np.linalg.norm(A @ e - rc) <= 1e-11 * (spla.norm(A) * np.linalg.norm(e))


def _scipy_vcycle(As, Us, lu, b, x, k=0):
    def gs(A, b, x, it):
        DL = sp.tril(A, 0).tocsr()
        x = x.copy()
        for _ in range(it):
            for c in range(x.shape[1]):
                x[:, c] += spla.spsolve_triangular(DL, b[:, c] - A @ x[:, c], lower=True)
        return x
    x = gs(As[k], b, x, 2)
    rc = Us[k].T @ (b - As[k] @ x)
    if k == len(Us) - 1:
        e = np.column_stack([lu.solve(rc[:, c]) for c in range(rc.shape[1])])
    else:
        e = _scipy_vcycle(As, Us, lu, rc, np.zeros_like(rc), k + 1)
    return gs(As[k], b, x + Us[k] @ e, 2)


def test_vcycle_and_solve_loop(P, oracle):
    """multiGridVCycleGS (:1059-1088) and the do-while solve loop (:1408-1419)."""
    As = [sp.csr_matrix(P.lhs)]
    for U in P.U:
        As.append(sp.csr_matrix(U.T @ As[-1] @ U))
    lu = spla.splu(sp.csc_matrix(As[-1]))
    O = oracle.Hierarchy(P.U, P.mass)
    O.set_system(P.lhs)
    b = P.rhs
    x1 = _scipy_vcycle(As, P.U, lu, b, b.copy())
    got = O.vcycle(b, b.copy())
    # forward error grows with the conditioning of the Poisson systems (||x||/||b|| ~ 1e8): compare backward
    assert np.linalg.norm(P.lhs @ (got - x1)) <= 1e-11 * spla.norm(P.lhs) * np.linalg.norm(x1)
    x, it, res, conv = O.solve(b, tol=1e-4, stop_type=2, max_iter=100)
    assert res <= 1e-4 and conv.shape == (it, 2) and np.all(np.diff(conv[:, 0]) >= 0)
    assert abs(oracle.residual_check(P.lhs, P.mass, b, x, 2) - res) <= 1e-6 * res + 1e-12
    # same count as the scipy iteration
    xs = b.copy(); its = 0
    while True:
        xs = _scipy_vcycle(As, P.U, lu, b, xs); its += 1
        r = P.lhs @ xs - b
        m = P.mass[:, None]
        if np.sqrt(((r ** 2 * m).sum(0) / (b ** 2 * m).sum(0)).max()) <= 1e-4 or its >= 100:
            break
    assert its == it
    # do-while: max_iter = 0 still runs one cycle; tol = 0 runs exactly max_iter cycles
    _, it1, _, _ = O.solve(b, tol=1.0, max_iter=0)
    assert it1 == 1
    _, it3, _, c3 = O.solve(b, tol=0.0, max_iter=3)
    assert it3 == 3 and c3.shape == (3, 2)
