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
                                dict(block_from_level=0), dict(sigma=1024, restrict_sigma=1024), dict(reorder_fine=1), dict(block_lanes=1), dict(block_lanes=1, block_ep=0), dict(block_lanes=1, block_csr=0)],
                         ids=["default", "lane1", "quad128", "exact", "blocked-all", "sorted-windows", "cluster-reorder", "entry-parallel", "block-csr", "lane1-sell64"])
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
        for which in ([0, 3, 4] + ([1, 2, 5, 6, 7] if blocked else [])):      # 5 = block-CSR off-block operator, 6 / 7 = lower / explicit parts of the unpadded block sweep (big blocked levels), else empty
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


@pytest.mark.parametrize("variant", ["default", "mixed", "device-coarse", "host-builders", "d3-blockcsr", "random-order", "jacobi"])
def test_same_pattern_refreshes_values_in_place(cabi, variant):
    """A system with the sparsity pattern of the live one (new tau) only moves values: LHS values up, numeric Galerkin
    passes, value refill of the layouts, numeric LDL^T (timing key setup_values_only).  Every array on the device and
    every iterate must equal what a fresh engine builds from scratch -- bitwise."""
    import scipy.sparse as sp
    big = variant == "d3-blockcsr"
    if big:
        P = problems.torus_problem(300, 280, "smoothing", 400)
    elif variant == "random-order":       # level 0 renumbered for locality (cluster order from the hierarchy)
        P = problems.torus_problem(160, 140, "smoothing", 100, order="random")
    else:
        P = problems.torus_problem(96, 80, "smoothing", 60)
    kw = {"mixed": dict(inner_precision=1), "device-coarse": dict(coarse_mode=cabi.COARSE_DEVICE_INVERSE),
          "host-builders": dict(device_setup=False), "d3-blockcsr": dict(block_lanes=1),
          "jacobi": dict(smoother=cabi.SMOOTHER_JACOBI)}.get(variant, {})
    lhs2 = (sp.diags(P.mass) + 7e-3 * P.S).tocsc()
    lhs3 = (sp.diags(P.mass) + 2e-4 * P.S).tocsr()          # CSR of a symmetric matrix: same arrays as its CSC
    eng = cabi.Engine(**kw)
    eng.set_prolongations(P.U); eng.set_mass(P.mass)
    eng.set_system(P.lhs)
    assert eng.timing("setup_values_only") == 0.0
    for lhs in (lhs2, lhs3, P.lhs):
        eng.set_system(lhs)
        assert eng.timing("setup_values_only") == (0.0 if variant == "host-builders" else 1.0)
        fresh = cabi.Engine(**kw)
        fresh.set_prolongations(P.U); fresh.set_mass(P.mass); fresh.set_system(lhs)
        for k in range(eng.num_levels):
            for which in (0, 1, 2, 5):
                try:
                    a, b = eng.debug_sell(k, which), fresh.debug_sell(k, which)
                except Exception:
                    continue
                if a is None or b is None:
                    assert a is None and b is None
                    continue
                for key in a:
                    if isinstance(a[key], np.ndarray):
                        assert np.array_equal(a[key], b[key]), (k, which, key)
        for k in range(eng.num_levels + 1):
            A1, A2 = eng.level_operator(k), fresh.level_operator(k)
            assert np.array_equal(A1.indptr, A2.indptr) and np.array_equal(A1.indices, A2.indices) and np.array_equal(A1.data, A2.data)
        eng.load_problem(P.rhs, P.rhs); fresh.load_problem(P.rhs, P.rhs)
        ra, rb = eng.run_cycles(4, 2), fresh.run_cycles(4, 2)
        assert np.array_equal(ra, rb)
        assert np.array_equal(eng.fetch_solution(), fresh.fetch_solution())
    # a changed pattern falls back to the full set-up
    E = sp.coo_matrix(([-1e-9, -1e-9], ([0, 777], [777, 0])), shape=lhs2.shape)
    eng.set_system(sp.csc_matrix(lhs2 + E))
    assert eng.timing("setup_values_only") == 0.0


def test_another_pattern_of_the_same_size_and_entry_count_takes_the_full_set_up(cabi):
    """gmg_set_system refreshes the live system's values AHEAD of the pattern verdict whenever size and entry count match (upload, Galerkin
    chain, refills and numeric LDL^T run beside the threads that inspect and digest the pattern).  When the verdict is "another pattern" --
    here: a few vertices renumbered, and then the live pattern in unsorted storage -- the full set-up must follow and give a fresh engine's bits."""
    import scipy.sparse as sp
    P = problems.torus_problem(96, 80, "smoothing", 60)
    n = P.lhs.shape[0]
    perm = np.arange(n); perm[100:400] = perm[100:400][::-1].copy()
    B = sp.csc_matrix(sp.csr_matrix(P.lhs)[perm][:, perm]); B.sort_indices()
    assert B.nnz == P.lhs.nnz and not (np.array_equal(B.indices, sp.csc_matrix(P.lhs).indices) and np.array_equal(B.indptr, sp.csc_matrix(P.lhs).indptr))
    eng = cabi.Engine(); eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    eng.set_system(B)
    assert eng.timing("setup_values_only") == 0.0
    fresh = cabi.Engine(); fresh.set_prolongations(P.U); fresh.set_mass(P.mass); fresh.set_system(B)
    for k in range(eng.num_levels + 1):
        A1, A2 = eng.level_operator(k), fresh.level_operator(k)
        assert np.array_equal(A1.indptr, A2.indptr) and np.array_equal(A1.indices, A2.indices) and np.array_equal(A1.data, A2.data)
    eng.load_problem(P.rhs, P.rhs); fresh.load_problem(P.rhs, P.rhs)
    assert np.array_equal(eng.run_cycles(4, 2), fresh.run_cycles(4, 2))
    assert np.array_equal(eng.fetch_solution(), fresh.fetch_solution())
    # back to the first pattern, then the same matrix with every column stored backwards: same size, same entry count, values in other places
    eng.set_system(P.lhs)
    A = sp.csc_matrix(P.lhs); A.sort_indices()
    idx, val = A.indices.copy(), A.data.copy()
    for j in range(n):
        idx[A.indptr[j]:A.indptr[j + 1]] = idx[A.indptr[j]:A.indptr[j + 1]][::-1]; val[A.indptr[j]:A.indptr[j + 1]] = val[A.indptr[j]:A.indptr[j + 1]][::-1]
    eng._chk(cabi.lib().gmg_set_system(eng._h, n, cabi._pi(A.indptr), cabi._pi(idx), cabi._pd(val)))
    assert eng.timing("setup_values_only") == 0.0
    fresh = cabi.Engine(); fresh.set_prolongations(P.U); fresh.set_mass(P.mass); fresh.set_system(A)
    for k in range(eng.num_levels + 1):
        A1, A2 = eng.level_operator(k), fresh.level_operator(k)
        assert np.array_equal(A1.indptr, A2.indptr) and np.array_equal(A1.indices, A2.indices) and np.array_equal(A1.data, A2.data)
    eng.load_problem(P.rhs, P.rhs); fresh.load_problem(P.rhs, P.rhs)
    assert np.array_equal(eng.run_cycles(4, 2), fresh.run_cycles(4, 2))
    # ... and a matrix of the live pattern whose values make the refresh fail (a zero on the diagonal) reports that, and the next good one works
    bad = sp.csc_matrix(P.lhs).copy(); bad.sort_indices()
    j = 1234; col = slice(bad.indptr[j], bad.indptr[j + 1]); bad.data[col] = np.where(bad.indices[col] == j, 0.0, bad.data[col])
    with pytest.raises(Exception):
        eng._chk(cabi.lib().gmg_set_system(eng._h, n, cabi._pi(bad.indptr), cabi._pi(bad.indices), cabi._pd(bad.data)))
    eng.set_system(P.lhs)
    ref = cabi.Engine(); ref.set_prolongations(P.U); ref.set_mass(P.mass); ref.set_system(P.lhs)
    eng.load_problem(P.rhs, P.rhs); ref.load_problem(P.rhs, P.rhs)
    assert np.array_equal(eng.run_cycles(4, 2), ref.run_cycles(4, 2))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sequences_of_systems_match_fresh_engines(cabi, seed):
    """A handle that lives through a random sequence of systems -- new values on the live pattern (the refresh that runs ahead of the pattern
    verdict), other patterns of the same size and entry count, other entry counts, unsorted storage, a matrix whose refresh fails, shorter and
    longer hierarchies, with and without the structure prepared from the point graph -- computes after every step what a fresh engine computes."""
    import scipy.sparse as sp
    from gravo_mg_amd import meshgen
    rng = np.random.default_rng(seed)
    P = problems.torus_problem(96, 80, "smoothing", 60)
    n = P.lhs.shape[0]
    base = sp.csc_matrix(P.lhs); base.sort_indices()
    S = P.S
    eng = cabi.Engine()
    state = {"U": P.U}
    graph = meshgen.neighbors_from_stiffness(S) if seed != 1 else None       # seeds 2, 3: the hierarchy announces the pattern (structure prepared)
    eng.set_prolongations(P.U, fine_graph=graph); eng.set_mass(P.mass)

    def fresh_check(lhs_canonical):
        fresh = cabi.Engine(); fresh.set_prolongations(state["U"]); fresh.set_mass(P.mass); fresh.set_system(lhs_canonical)
        eng.load_problem(P.rhs, P.rhs); fresh.load_problem(P.rhs, P.rhs)
        assert np.array_equal(eng.run_cycles(2, 2), fresh.run_cycles(2, 2))
        assert np.array_equal(eng.fetch_solution(), fresh.fetch_solution())
        fresh.close()

    eng.set_system(base); fresh_check(base)
    for step in range(14):
        op = int(rng.integers(0, 6))
        tau = float(10.0 ** rng.uniform(-4, -2))
        A = sp.csc_matrix(sp.diags(P.mass) + tau * S); A.sort_indices()
        if op == 0:                                        # the live pattern, new values
            eng.set_system(A); fresh_check(A)
        elif op == 1:                                      # a few vertices renumbered: same size, same entry count, another pattern
            lo = int(rng.integers(0, n - 400)); perm = np.arange(n); perm[lo:lo + 300] = perm[lo:lo + 300][::-1].copy()
            B = sp.csc_matrix(sp.csr_matrix(A)[perm][:, perm]); B.sort_indices()
            eng.set_system(B); fresh_check(B)
        elif op == 2:                                      # one more symmetric coupling
            i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
            if i == j:
                continue
            B = sp.csc_matrix(A + sp.coo_matrix(([-1e-9, -1e-9], ([i, j], [j, i])), shape=A.shape)); B.sort_indices()
            eng.set_system(B); fresh_check(B)
        elif op == 3:                                      # every column stored backwards
            idx, val = A.indices.copy(), A.data.copy()
            for c in range(0, n, 3):
                idx[A.indptr[c]:A.indptr[c + 1]] = idx[A.indptr[c]:A.indptr[c + 1]][::-1]; val[A.indptr[c]:A.indptr[c + 1]] = val[A.indptr[c]:A.indptr[c + 1]][::-1]
            eng._chk(cabi.lib().gmg_set_system(eng._h, n, cabi._pi(A.indptr), cabi._pi(idx), cabi._pd(val)))
            fresh_check(A)
        elif op == 4:                                      # a zero on the diagonal: an error, and no system afterwards
            bad = A.copy(); c = int(rng.integers(0, n)); col = slice(bad.indptr[c], bad.indptr[c + 1])
            bad.data[col] = np.where(bad.indices[col] == c, 0.0, bad.data[col])
            with pytest.raises(Exception):
                eng._chk(cabi.lib().gmg_set_system(eng._h, n, cabi._pi(bad.indptr), cabi._pi(bad.indices), cabi._pd(bad.data)))
            eng.set_system(A); fresh_check(A)
        else:                                              # another hierarchy depth on the same handle
            state["U"] = P.U[:1] if len(state["U"]) == len(P.U) else P.U
            eng.set_prolongations(state["U"], fine_graph=graph)
            eng.set_system(A); fresh_check(A)


def test_non_canonical_lhs_storage_is_accepted(cabi):
    """Unsorted row indices and duplicate entries (summed, like Eigen's setFromTriplets) give the same system."""
    import scipy.sparse as sp
    P = problems.torus_problem(48, 40, "poisson", 60)
    A = sp.csc_matrix(P.lhs); A.sort_indices()
    rng = np.random.default_rng(5)
    ptr, idx, val = [0], [], []
    for j in range(A.shape[0]):
        r = A.indices[A.indptr[j]:A.indptr[j + 1]].copy(); v = A.data[A.indptr[j]:A.indptr[j + 1]].copy()
        if j % 3 == 0:                 # split the first entry in two halves (a duplicate)
            r = np.concatenate([r, r[:1]]); v = np.concatenate([v, v[:1] * 0.5]); v[0] *= 0.5
        perm = rng.permutation(len(r))
        idx.append(r[perm]); val.append(v[perm]); ptr.append(ptr[-1] + len(r))
    messy = sp.csc_matrix((np.concatenate(val), np.concatenate(idx).astype(np.int32), np.array(ptr, np.int32)), shape=A.shape)
    assert not messy.has_canonical_format
    ref = cabi.Engine(); ref.set_prolongations(P.U); ref.set_mass(P.mass); ref.set_system(A)
    eng = cabi.Engine(); eng.set_prolongations(P.U); eng.set_mass(P.mass)
    eng._chk(cabi.lib().gmg_set_system(eng._h, A.shape[0], cabi._pi(messy.indptr), cabi._pi(messy.indices), cabi._pd(messy.data)))
    eng._n0 = A.shape[0]
    assert abs(eng.level_operator(0) - A).max() <= 1e-15 * abs(A).max()
    ref.load_problem(P.rhs, P.rhs); eng.load_problem(P.rhs, P.rhs)
    assert np.allclose(eng.run_cycles(3, 2), ref.run_cycles(3, 2), rtol=1e-9)


def test_level_operators_are_fetched_on_demand_and_repeat_systems_reuse_the_pattern(cabi):
    """Second set_system with the same pattern (ordering cache hit: no host copies of the middle levels are made) must
    give the same operators and iterates as the first."""
    P = problems.torus_problem(96, 80, "poisson", 30)
    eng = cabi.Engine(); eng.set_prolongations(P.U); eng.set_mass(P.mass)
    eng.set_system(P.lhs)
    first = [eng.level_operator(k) for k in range(eng.num_levels + 1)]
    eng.load_problem(P.rhs, P.rhs); h1 = eng.run_cycles(3, 2)
    eng.set_system(P.lhs)
    assert eng.timing("setup_ordering_cached") == 1.0
    eng.load_problem(P.rhs, P.rhs); h2 = eng.run_cycles(3, 2)
    assert np.array_equal(h1, h2)
    for k, a in enumerate(first):
        b = eng.level_operator(k)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data), k
    # a different matrix with the same pattern
    lhs2 = P.lhs.copy(); lhs2.data *= 1.5
    eng.set_system(lhs2)
    assert eng.timing("setup_ordering_cached") == 1.0
    assert np.allclose(eng.level_operator(eng.num_levels).toarray(), 1.5 * first[-1].toarray(), rtol=1e-12)
    x, it, res, _ = eng.solve(1.5 * P.rhs, tol=1e-6)
    assert res <= 1e-6


def _star_coupled(P, hub, n_links, rng):
    """lhs + sum_j w (e_hub - e_j)(e_hub - e_j)^T: one very long row / column (SPD is preserved)."""
    import scipy.sparse as sp
    n = P.lhs.shape[0]
    others = rng.choice(np.setdiff1d(np.arange(n), [hub]), n_links, replace=False)
    w = 1e-3 * abs(P.lhs.diagonal()).mean()
    rows = np.concatenate([np.full(n_links, hub), others, np.full(n_links, hub), others])
    cols = np.concatenate([np.full(n_links, hub), others, others, np.full(n_links, hub)])
    vals = np.concatenate([np.full(2 * n_links, w), np.full(2 * n_links, -w)])
    return sp.csc_matrix(P.lhs + sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsc())


@pytest.mark.parametrize("n_links", [150, 600], ids=["long-row", "long-row-and-rap-overflow"])
def test_inputs_the_device_builders_cannot_take_fall_back_to_the_host_builders(cabi, oracle, n_links):
    """A row longer than the device SELL builder's private buffer (96) and a coarse row with more distinct columns than
    the device RAP's hash set (256): same operators and iterates as the host-planned engine, and the solve converges."""
    P = problems.torus_problem(96, 80, "poisson", 30)
    lhs = _star_coupled(P, 1234, n_links, np.random.default_rng(3))
    dev, host = [cabi.Engine(device_setup=d) for d in (True, False)]
    for e in (dev, host):
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(lhs)
    O = oracle.Hierarchy(P.U, P.mass); O.set_system(lhs)
    for k in range(dev.num_levels + 1):
        a, b = dev.level_operator(k), host.level_operator(k)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data), k
        A = O.level_operator(k)
        assert abs(a - A).max() <= 1e-12 * abs(A).max()
    dev.load_problem(P.rhs, P.rhs); host.load_problem(P.rhs, P.rhs)
    assert np.array_equal(dev.run_cycles(3, 2), host.run_cycles(3, 2))
    x, it, res, _ = dev.solve(P.rhs, tol=1e-6)
    assert res <= 1e-6 and it < 60
    assert abs(oracle.residual_check(lhs, P.mass, P.rhs, x, 2) - res) <= 1e-3 * res + 1e-7


def test_prolongation_rows_with_more_than_three_entries_use_the_host_builders(cabi, oracle):
    """The device RAP / layout builder assume <= 3 entries per prolongation row (what the reference builds); other
    hierarchies go through the host builders with identical semantics."""
    import scipy.sparse as sp
    P = problems.torus_problem(64, 60, "smoothing", 60)     # well conditioned: far-away prolongation entries do not stall it
    U0 = sp.lil_matrix(P.U[0])
    rng = np.random.default_rng(11)
    for i in rng.choice(U0.shape[0], 40, replace=False):
        free = np.setdiff1d(np.arange(U0.shape[1]), U0.rows[i])
        for j in rng.choice(free, 2, replace=False):
            U0[i, j] = 0.01
    Us = [sp.csc_matrix(U0)] + list(P.U[1:])
    assert (np.diff(sp.csr_matrix(Us[0]).indptr) > 3).any()
    eng = cabi.Engine(); eng.set_prolongations(Us); eng.set_mass(P.mass)
    for rep in range(2):            # the second call reuses the cached orderings and the flagged device transfers
        eng.set_system(P.lhs)
        O = oracle.Hierarchy(Us, P.mass); O.set_system(P.lhs)
        for k in range(eng.num_levels + 1):
            A = O.level_operator(k)
            assert abs(eng.level_operator(k) - A).max() <= 1e-12 * abs(A).max(), (rep, k)
        r = np.random.default_rng(1).standard_normal((P.n, 1))
        assert rel(eng.restrict(0, r), Us[0].T @ r) <= 1e-13
        x, it, res, _ = eng.solve(P.rhs, tol=1e-6)
        xo, ito, reso, _ = O.solve(P.rhs, tol=1e-6)
        assert res <= 1e-6 and reso <= 1e-6 and abs(it - ito) <= 2


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def test_base_order_of_a_reordered_level_follows_the_gather_score(cabi):
    """Inputs without locality get their finest level renumbered; with the hierarchy object's breadth-first order at hand the
    engine scores it against the cluster order on the actual matrix (distinct cache lines per gather of 64 rows) and takes the
    better one: breadth-first on a triangle mesh, cluster order on a kNN point cloud.  Device score and host twin decide alike
    (same orderings from the device builder and the host planner), and the solve agrees with the oracle either way."""
    from gravo_mg_amd import meshgen
    from oracle import oracle
    V, F = meshgen.torus_mesh(340, 340, order="random")
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=2000)
    lhs, rhs = meshgen.poisson_system(S, mass)
    assert H.fine_order is not None
    engs = []
    for dev in (True, False):
        e = cabi.Engine(device_setup=dev)
        e.use_hierarchy(H); e.set_mass(mass); e.set_system(lhs)
        assert e.timing("base_order_choice") == 1.0 and 0 < e.timing("base_order_score_bfs") < e.timing("base_order_score_cluster")
        engs.append(e)
    assert engs[0].timing("base_order_score_bfs") == engs[1].timing("base_order_score_bfs")
    assert engs[0].timing("base_order_score_cluster") == engs[1].timing("base_order_score_cluster")
    for a, b in zip(engs[0].level_ordering(0), engs[1].level_ordering(0)):
        assert np.array_equal(a, b)
    x, it, res, _ = engs[0].solve(rhs, tol=1e-4)
    O = oracle.Hierarchy(H.U, mass); O.set_system(lhs)
    xo, ito, reso, _ = O.solve(rhs, tol=1e-4)
    assert res <= 1e-4 and it <= ito + 1
    assert np.sqrt((mass[:, None] * (x - xo) ** 2).sum() / (mass[:, None] * xo ** 2).sum()) <= 20 * 1e-4
    n_col_bfs = engs[0].level_info(0)["n_colors"]
    # the cluster order alone (no fine order handed over): more colour classes on the same mesh
    e2 = cabi.Engine(); e2.set_prolongations(H.U); e2.set_mass(mass); e2.set_system(lhs)
    assert e2.timing("base_order_choice") == 0.0 and n_col_bfs <= e2.level_info(0)["n_colors"]
    x2, it2, res2, _ = e2.solve(rhs, tol=1e-4)
    assert res2 <= 1e-4 and np.sqrt((mass[:, None] * (x - x2) ** 2).sum() / (mass[:, None] * x2 ** 2).sum()) <= 20 * 1e-4
    # a kNN point cloud: thick, ragged wavefronts -- the cluster order scores better
    P = meshgen.torus_points(120_000, noise=0.0005)
    Sp, mp = meshgen.knn_graph_laplacian(P, 8)
    Hp = cabi.Hierarchy(P, meshgen.neighbors_from_stiffness(Sp), lower_bound=2000)
    assert Hp.fine_order is not None
    ep = cabi.Engine(block_fine=0); ep.use_hierarchy(Hp); ep.set_mass(mp); ep.set_system(meshgen.poisson_system(Sp, mp)[0])      # (colour-major level 0: a blocked one takes its blocks from the cluster order and is not renumbered)
    assert ep.timing("base_order_choice") == 0.0 and ep.timing("base_order_score_cluster") < ep.timing("base_order_score_bfs")


@pytest.mark.parametrize("case,test_fail", [("torus", 0), ("random-order", 0), ("pointcloud", 0), ("smoothing-d3", 0), ("torus", 9), ("pointcloud", 11),
                                            ("smoothing-d3", 10), ("torus", -3), ("random-order", -2)])
def test_16_bit_column_codes_of_the_fine_level_change_nothing_but_the_bytes(cabi, case, test_fail):
    """Level 0 stores its column indices a second time as 16-bit codes (window of the slice + offset inside it, gmgs::compress_cols)
    and the fine-level kernels read those: the same columns in the same order, so every iterate is bit-identical to the 32-bit path
    (gmg_config::fine_col16 = 0), also after a values-only refresh, for d = 3 and in the fp32 inner cycle."""
    import os
    P = {"torus": lambda: problems.torus_problem(96, 80, "poisson", 30),
         "random-order": lambda: problems.torus_problem(64, 60, "poisson", 40, order="random"),
         "pointcloud": lambda: problems.pointcloud_problem(3000),
         "smoothing-d3": lambda: problems.torus_problem(64, 60, "smoothing", 60)}[case]()

    def run(no16):
        e = cabi.Engine(fine_col16=not no16)
        if not no16 and test_fail:               # pretend slices are not covered by their windows: every N-th (flagged one by one), or the first -N (prefix)
            e.debug_set("col16_uncovered", test_fail)
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
        return e
    a, b = run(False), run(True)
    assert b.timing("col16_l0") == 0.0
    assert b.timing("col16_R_l0") == 0.0 and b.timing("col16_P_l0") == 0.0
    for key in ("col16_l0", "col16_R_l0", "col16_P_l0"):               # the operator and both transfers of level 0
        assert a.timing(key) == 1.0
        if test_fail == 0:
            assert a.timing(key + "_failed_slices") == 0.0 and a.timing(key + "_mode") == 1.0     # small meshes: 8 windows of 8 192 cover any slice
        elif test_fail > 0:                                            # uncovered slices all over the numbering: found through their flags
            assert a.timing(key + "_failed_slices") > 0 and a.timing(key + "_mode") == 2.0
        else:                                                          # a short prefix of uncovered slices: codes from the slice behind it
            assert a.timing(key + "_failed_slices") == -test_fail and a.timing(key + "_mode") == 1.0
    for e in (a, b):
        e.load_problem(P.rhs, P.rhs)
    ha, hb = a.run_cycles(4, 2), b.run_cycles(4, 2)
    assert np.array_equal(ha, hb) and np.array_equal(a.fetch_solution(), b.fetch_solution())
    rng = np.random.default_rng(5)
    x = rng.standard_normal(P.rhs.shape)
    assert np.array_equal(a.residual(0, P.rhs, x), b.residual(0, P.rhs, x)) and np.array_equal(a.smooth(0, P.rhs, x, 2), b.smooth(0, P.rhs, x, 2))
    for t in (0, 1, 2, 3):
        assert a.residual_norm(P.rhs, x, t) == b.residual_norm(P.rhs, x, t)
    lhs2 = P.lhs.copy(); lhs2.data = lhs2.data * (1.0 + 0.1 * rng.random(lhs2.nnz))
    lhs2 = (lhs2 + lhs2.T) * 0.5                                        # same pattern, new (symmetric) values: refreshed in place
    a.set_system(lhs2); b.set_system(lhs2)
    assert a.timing("setup_values_only") == 1.0
    assert np.array_equal(a.smooth(0, P.rhs, x, 2), b.smooth(0, P.rhs, x, 2))


@pytest.mark.parametrize("case,test_fail", [("torus", 0), ("torus", 9), ("torus", -3), ("smoothing-d3", 0), ("random-order", 0), ("pointcloud", 0), ("torus-mixed", 0)])
def test_equally_wide_slices_are_read_without_their_pointers(cabi, case, test_fail):
    """gmg_config::uniform_slices: when every 64-row slice of a level-0 operator is equally wide (a regular mesh: six neighbours everywhere; the
    prolongation: three entries per row) the kernels compute a slice's place from its number instead of loading two slice pointers first
    (kernels.hip.hpp::row_dot).  Found by gmgs::compress_cols; same entries in the same order, so every iterate is bit-identical -- with the codes
    flagged slice by slice and with a 32-bit prefix too --; an operator with slices of several widths (a kNN graph) keeps its pointers."""
    P = {"torus": lambda: problems.torus_problem(96, 80, "poisson", 30),
         "random-order": lambda: problems.torus_problem(64, 60, "poisson", 40, order="random"),
         "pointcloud": lambda: problems.pointcloud_problem(3000),
         "torus-mixed": lambda: problems.torus_problem(96, 80, "poisson", 30),          # the fp32 inner cycle reads the same layout through its fp32 twins
         "smoothing-d3": lambda: problems.torus_problem(64, 60, "smoothing", 60)}[case]()

    def run(uniform):
        e = cabi.Engine(uniform_slices=uniform, block_fine=0, **({"inner_precision": 1} if case.endswith("mixed") else {}))
        if test_fail:
            e.debug_set("col16_uncovered", test_fail)
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
        return e
    a, b = run(True), run(False)
    for key in ("col16_l0", "col16_R_l0", "col16_P_l0"):
        assert b.timing(key + "_uniform_width") == 0.0
    if case == "pointcloud":
        assert a.timing("col16_l0_uniform_width") == 0.0                # 8 neighbours at least, more where the relation is not mutual
    else:
        assert a.timing("col16_l0_uniform_width") == 6.0                # a closed triangle mesh of a torus: valence six
    for e in (a, b):
        e.load_problem(P.rhs, P.rhs)
    ha, hb = a.run_cycles(4, 2), b.run_cycles(4, 2)
    assert np.array_equal(ha, hb) and np.array_equal(a.fetch_solution(), b.fetch_solution())
    rng = np.random.default_rng(5)
    x = rng.standard_normal(P.rhs.shape)
    assert np.array_equal(a.residual(0, P.rhs, x), b.residual(0, P.rhs, x)) and np.array_equal(a.smooth(0, P.rhs, x, 2), b.smooth(0, P.rhs, x, 2))
    for t in (0, 1, 2, 3):
        assert a.residual_norm(P.rhs, x, t) == b.residual_norm(P.rhs, x, t)
    r = rng.standard_normal(P.rhs.shape)
    assert np.array_equal(a.restrict(0, r), b.restrict(0, r))
    e1 = rng.standard_normal((a.level_info(1)["n"], P.rhs.shape[1]))
    assert np.array_equal(a.prolong_add(0, e1, x), b.prolong_add(0, e1, x))
    lhs2 = P.lhs.copy(); lhs2.data = lhs2.data * (1.0 + 0.1 * rng.random(lhs2.nnz))
    lhs2 = (lhs2 + lhs2.T) * 0.5
    a.set_system(lhs2); b.set_system(lhs2)
    assert a.timing("setup_values_only") == 1.0
    assert np.array_equal(a.smooth(0, P.rhs, x, 2), b.smooth(0, P.rhs, x, 2))


@pytest.mark.parametrize("case", ["torus", "random-order", "pointcloud", "smoothing-d3-mixed"])
def test_structure_prepared_at_hierarchy_time_gives_the_cold_set_ups_bits(cabi, case):
    """gmg_set_fine_graph / gmg_use_hierarchy hand the engine the hierarchy's point graph -- the sparsity pattern of the systems it is built
    for -- and gmg_finalize_hierarchy prepares the structure on placeholder values (gmg_config::prepare_structure): the first gmg_set_system
    is then a values-only refresh.  It must leave exactly what a cold set-up leaves: same orderings, same iterates, bit for bit.  A handle
    in the placeholder state has no system (solves are refused), and a system with another pattern takes the cold path."""
    from gravo_mg_amd import meshgen
    if case == "pointcloud":
        pos = meshgen.torus_points(6000, noise=0.002)
        S, mass = meshgen.knn_graph_laplacian(pos, 8)
        lhs, rhs = meshgen.poisson_system(S, mass)
        lb, kw = 150, {}
    else:
        pos, F = meshgen.torus_mesh(96, 80, order="random" if case == "random-order" else "natural")
        S, mass = meshgen.cotan_laplacian(pos, F)
        lhs, rhs = meshgen.smoothing_system(S, mass, pos) if case.startswith("smoothing") else meshgen.poisson_system(S, mass)
        lb, kw = 60, ({"inner_precision": 1} if case.endswith("mixed") else {})
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(pos, neigh, lower_bound=lb)

    prepared = cabi.Engine(**kw)
    prepared.use_hierarchy(H)
    assert prepared.timing("structure_prepare_ms") > 0.0
    with pytest.raises(cabi.GmgError):                      # placeholder values: nothing to solve with
        prepared.solve(rhs)
    prepared.set_mass(mass); prepared.set_system(lhs)
    assert prepared.timing("setup_values_only") == 1.0 and prepared.timing("setup_structure_prepared") == 1.0

    table = cabi.Engine(**kw)                                # the same through the table (what the C++ mirror does: it keeps `neigh`, not the hierarchy object)
    table.set_prolongations(H.U, fine_order=H.fine_order, fine_graph=neigh)
    table.set_mass(mass); table.set_system(lhs)
    assert table.timing("setup_structure_prepared") == 1.0

    cold = cabi.Engine(prepare_structure=False, **kw)
    cold.use_hierarchy(H); cold.set_mass(mass); cold.set_system(lhs)
    assert cold.timing("setup_values_only") == 0.0 and cold.timing("setup_structure_prepared") == 0.0

    for e in (prepared, table):
        for k in range(cold.num_levels):
            a, b = e.level_ordering(k), cold.level_ordering(k)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        xa, ita, resa, conva = e.solve(rhs, tol=1e-6, max_iter=60)
        xb, itb, resb, convb = cold.solve(rhs, tol=1e-6, max_iter=60)
        assert ita == itb and np.array_equal(conva[:, 1], convb[:, 1]) and np.array_equal(xa, xb)
        assert e.residual_norm(rhs, xa, 2) == cold.residual_norm(rhs, xb, 2)
    # another pattern (the Bilaplacian's two-ring): cold path on the prepared handle, same bits as on a fresh one
    if case == "torus":
        lhs2, rhs2 = meshgen.smoothing_system(meshgen.bilaplacian(S, mass), mass, pos[:, :1], tau=1e-9)
        prepared.set_system(lhs2)
        assert prepared.timing("setup_values_only") == 0.0 and prepared.timing("setup_structure_prepared") == 0.0
        fresh = cabi.Engine(prepare_structure=False)
        fresh.use_hierarchy(H); fresh.set_mass(mass); fresh.set_system(lhs2)
        xa = prepared.solve(rhs2, tol=1e-3, max_iter=8)[0]
        xb = fresh.solve(rhs2, tol=1e-3, max_iter=8)[0]
        assert np.array_equal(xa, xb)


def test_an_asymmetric_point_graph_is_not_prepared_for(cabi):
    """The structure is prepared for systems with the point graph's pattern (tau M + S of a symmetric graph).  A kNN table that is not symmetric --
    j among i's neighbours, i not among j's -- cannot be the pattern of a symmetric system: gmg_finalize_hierarchy notices (one threaded pass) and
    skips the placeholder set-up instead of paying for it and taking the cold path anyway (round-5 advice); results are those of a cold handle."""
    from gravo_mg_amd import meshgen
    pos = meshgen.torus_points(6000, noise=0.002)
    S, mass = meshgen.knn_graph_laplacian(pos, 8)
    lhs, rhs = meshgen.poisson_system(S, mass)
    neigh = meshgen.neighbors_from_stiffness(S).copy()
    i = 123
    j = int([v for v in neigh[i] if v >= 0 and v != i][0])
    row = neigh[j]
    assert i in row and j in neigh[i]
    row[row == i] = [v for v in row if v >= 0 and v != i and v != j][0]      # j forgets i (the slot repeats another neighbour): i -> j stays
    H = cabi.Hierarchy(pos, neigh, lower_bound=150)
    e = cabi.Engine()
    e.set_prolongations(H.U, fine_order=H.fine_order, fine_graph=neigh)       # the table as the caller holds it (gmg_set_fine_graph: what the C++ mirror passes)
    assert e.timing("structure_prepare_symmetric_graph") == 0.0
    e.set_mass(mass); e.set_system(lhs)
    assert e.timing("setup_structure_prepared") == 0.0 and e.timing("setup_values_only") == 0.0
    cold = cabi.Engine(prepare_structure=False)
    cold.use_hierarchy(H); cold.set_mass(mass); cold.set_system(lhs)
    xa, ita, resa, _ = e.solve(rhs, tol=1e-6, max_iter=60)
    xb, itb, resb, _ = cold.solve(rhs, tol=1e-6, max_iter=60)
    assert ita == itb and np.array_equal(xa, xb)
