"""The device-side layout builder (setup_kernels.hip.hpp) against its specification, the host planner
(host_plan.hpp): every SELL array of every level must come out bit-identical, for colour-major and blocked
levels, both lane layouts, d = 1 and 3, natural and random vertex order, mesh and point-cloud matrices."""
import numpy as np
import pytest

from tests import problems

pytestmark = pytest.mark.gpu


def _engines(cabi, P, **kw):
    out = []
    for dev in (True, False):
        e = cabi.Engine(device_setup=dev, **kw)
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
        out.append(e)
    return out


@pytest.mark.parametrize("case", ["torus-L3", "random-order", "pointcloud", "smoothing", "bilaplacian"])
@pytest.mark.parametrize("kw", [dict(), dict(block_lanes=1, block_rows=256), dict(block_lanes=4, block_rows=128), dict(block_rows=0),
                                dict(block_from_level=0), dict(sigma=0)], ids=["default", "lane1", "quad128", "exact", "blocked-all", "nosort"])
def test_device_layout_equals_host_layout(cabi, case, kw):
    P = {"torus-L3": lambda: problems.torus_problem(96, 80, "poisson", 30),
         "random-order": lambda: problems.torus_problem(48, 40, "poisson", 40, order="random"),
         "pointcloud": lambda: problems.pointcloud_problem(3000),
         "smoothing": lambda: problems.torus_problem(64, 60, "smoothing", 60),
         "bilaplacian": lambda: problems.torus_problem(48, 40, "bilaplacian", 40)}[case]()
    dev, host = _engines(cabi, P, **kw)
    assert dev.num_levels == host.num_levels
    for k in range(dev.num_levels):
        nd, cd = dev.level_ordering(k); nh, ch = host.level_ordering(k)
        assert np.array_equal(nd, nh) and np.array_equal(cd, ch)
        blocked = dev.level_blocks(k) is not None
        for which in ([0, 3, 4] + ([1, 2] if blocked else [])):
            a, b = dev.debug_sell(k, which), host.debug_sell(k, which)
            assert (a["n_slices"], a["lpr"]) == (b["n_slices"], b["lpr"]), (k, which)
            assert np.array_equal(a["slice_ptr"], b["slice_ptr"]), (k, which)
            assert np.array_equal(a["col"], b["col"]), (k, which)
            assert np.array_equal(a["val"], b["val"]), (k, which)           # bitwise
            if b["row_of"] is not None:
                assert np.array_equal(a["row_of"], b["row_of"]), (k, which)
            else:
                assert a["row_of"] is None
            if which == 0:
                assert np.array_equal(a["diag"], b["diag"])
    # and the two engines produce identical iterates
    dev.load_problem(P.rhs, P.rhs); host.load_problem(P.rhs, P.rhs)
    assert np.array_equal(dev.run_cycles(3, 2), host.run_cycles(3, 2))
    assert np.array_equal(dev.fetch_solution(), host.fetch_solution())


def test_device_builder_reports_missing_diagonal(cabi):
    import scipy.sparse as sp
    P = problems.torus_problem(48, 40, "poisson", 60)
    eng = cabi.Engine(device_setup=True)
    eng.set_prolongations(P.U)
    bad = P.lhs.tolil(); bad[7, 7] = 0.0
    bad = sp.csc_matrix(bad); bad.eliminate_zeros()
    with pytest.raises(cabi.GmgError) as ei:
        eng.set_system(bad)
    assert ei.value.code == cabi.GMG_ERR_NUMERIC


def test_long_rows_fall_back_to_the_host_planner(cabi):
    """Rows longer than the device builder's private sort buffer (96 entries) -> host planner, same results."""
    import scipy.sparse as sp
    P = problems.torus_problem(48, 40, "poisson", 60)
    n = P.n
    rng = np.random.default_rng(0)
    # densify a few rows/columns symmetrically with tiny couplings (keeps the matrix SPD-ish and diagonally dominant)
    rows = np.repeat(np.array([3, 500, 1200]), 150)
    cols = rng.choice(n, size=rows.size, replace=False) if rows.size <= n else rng.integers(0, n, rows.size)
    E = sp.coo_matrix((np.full(rows.size, -1e-9), (rows, cols)), shape=(n, n)).tocsr()
    E = E + E.T
    lhs = sp.csc_matrix(P.lhs + E + sp.diags(np.asarray(abs(E).sum(axis=1)).ravel()))
    dev, host = [cabi.Engine(device_setup=d) for d in (True, False)]
    for e in (dev, host):
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(lhs)
        e.load_problem(P.rhs, P.rhs)
    assert np.array_equal(dev.run_cycles(2, 2), host.run_cycles(2, 2))


@pytest.mark.parametrize("case", ["torus-L3", "random-order", "pointcloud", "smoothing", "bilaplacian"])
def test_device_galerkin_is_bitwise_the_host_galerkin(cabi, case):
    """gmgs::rap_rows walks the (i, j, c) triples in the host implementation's order with separately rounded
    multiply and add: A_k (k >= 1) must be bit-identical, pattern included (multigrid_solver.cpp:1387-1392)."""
    P = {"torus-L3": lambda: problems.torus_problem(96, 80, "poisson", 30),
         "random-order": lambda: problems.torus_problem(48, 40, "poisson", 40, order="random"),
         "pointcloud": lambda: problems.pointcloud_problem(3000),
         "smoothing": lambda: problems.torus_problem(64, 60, "smoothing", 60),
         "bilaplacian": lambda: problems.torus_problem(48, 40, "bilaplacian", 40)}[case]()
    engs = []
    for dev in (True, False):
        e = cabi.Engine(device_rap=dev)
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
        engs.append(e)
    for k in range(1, engs[0].num_levels + 1):
        a, b = engs[0].level_operator(k), engs[1].level_operator(k)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
        assert np.array_equal(a.data, b.data)
        assert a.has_sorted_indices
    assert engs[0].timing("reduction") > 0


def test_ordering_cache_reused_for_same_pattern(cabi):
    """demos/smoothing.py re-solves with a new tau: same sparsity pattern, new values.  The second gmg_set_system reuses
    the orderings (and must give exactly what a fresh engine gives); a different pattern or hierarchy must not."""
    import scipy.sparse as sp
    from gravo_mg_amd import meshgen
    P = problems.torus_problem(64, 60, "smoothing", 60)
    lhs2 = (sp.diags(P.mass) + 5e-3 * P.S).tocsc()
    eng = cabi.Engine()
    eng.set_prolongations(P.U); eng.set_mass(P.mass)
    eng.set_system(P.lhs)
    assert eng.timing("setup_ordering_cached") == 0.0
    eng.set_system(lhs2)
    assert eng.timing("setup_ordering_cached") == 1.0
    fresh = cabi.Engine()
    fresh.set_prolongations(P.U); fresh.set_mass(P.mass); fresh.set_system(lhs2)
    xa, ita, resa, _ = eng.solve(P.rhs, tol=1e-8)
    xb, itb, resb, _ = fresh.solve(P.rhs, tol=1e-8)
    assert ita == itb and np.array_equal(xa, xb)
    # a different pattern (one extra symmetric coupling) misses the cache
    E = sp.coo_matrix(([-1e-9, -1e-9], ([0, 777], [777, 0])), shape=lhs2.shape)
    eng.set_system(sp.csc_matrix(lhs2 + E))
    assert eng.timing("setup_ordering_cached") == 0.0
    # a new hierarchy invalidates it as well
    eng.set_prolongations(P.U[:1])
    eng.set_system(lhs2)
    assert eng.timing("setup_ordering_cached") == 0.0
