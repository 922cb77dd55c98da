"""Host-side logic of libgravomg_hip.so that needs no GPU: the hierarchy builder, the Galerkin product, the
device-layout planner, the coarsest-level LDL^T, the C-ABI surface and its error behaviour on a box without
a HIP device.  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from tests import problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------- C-ABI surface
def test_library_exports_every_declared_symbol(cabi):
    """Every function include/gravomg_hip.h (the drop-in boundary) and include/gravomg_hip_internal.h (test / measurement hooks) declare
    is exported by the built library and typed in cabi.SIGNATURES / cabi.INTERNAL_SIGNATURES; the boundary stays at <= 75 entry points."""
    lib = C.CDLL(cabi.LIB_PATH)
    for header, table in (("gravomg_hip.h", cabi.SIGNATURES), ("gravomg_hip_internal.h", cabi.INTERNAL_SIGNATURES)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        declared = set(re.findall(r"\b(gmg_[a-z0-9_]+)\s*\(", hdr))
        for name in declared:
            assert hasattr(lib, name), f"{name} declared in {header} but not exported"
        assert declared == set(table), (header, declared ^ set(table))
    assert 40 <= len(cabi.SIGNATURES) <= 75


def test_config_defaults_follow_reference_python_defaults(cabi):
    cfg = cabi.GmgConfig()
    assert cabi.lib().gmg_config_default(C.byref(cfg)) == 0
    # gravomg_bindings/src/gravomg/core.py:10  pre_iters=2, post_iters=2
    assert (cfg.pre_iters, cfg.post_iters) == (2, 2)
    assert cfg.smoother == cabi.SMOOTHER_MULTICOLOR_GS and cfg.coarse_mode == cabi.COARSE_AUTO
    opt = cabi.GmgHierarchyOptions()
    assert cabi.lib().gmg_hierarchy_options_default(C.byref(opt)) == 0
    assert (opt.ratio, opt.lower_bound, opt.check_voronoi, opt.nested, opt.sampling, opt.weighting) == (8.0, 1000, 1, 0, 0, 0)


def test_invalid_config_rejected(cabi):
    for kw in (dict(sigma=100), dict(row_align=32), dict(block_rows=2048), dict(block_rows=100), dict(pre_iters=-1)):
        with pytest.raises(cabi.GmgError):
            cabi.Engine(**kw)


def test_device_entry_points_fail_loudly_without_a_gpu(cabi):
    """No CPU fallback behind the C-ABI: on a box without a HIP device every device call is GMG_ERR_NO_DEVICE."""
    if cabi.device_count() > 0:
        pytest.skip("a HIP device is present")
    P = problems.torus_problem(24, 20, "poisson", 20)
    eng = cabi.Engine()
    eng.set_prolongations(P.U)          # host-side state is accepted ...
    eng.set_mass(P.mass)
    with pytest.raises(cabi.GmgError) as ei:
        eng.set_system(P.lhs)           # ... anything that needs the device is refused
    assert ei.value.code == cabi.GMG_ERR_NO_DEVICE and "no CPU fallback" in str(ei.value)
    for call in (lambda: eng.vcycle(P.rhs, P.rhs), lambda: eng.solve(P.rhs), lambda: eng.smooth(0, P.rhs, P.rhs, 1),
                 lambda: eng.residual_norm(P.rhs, P.rhs), lambda: eng.load_problem(P.rhs, P.rhs), lambda: eng.run_cycles(1)):
        with pytest.raises(cabi.GmgError) as ei:
            call()
        assert ei.value.code == cabi.GMG_ERR_NO_DEVICE


def test_argument_validation(cabi):
    P = problems.torus_problem(24, 20, "poisson", 20)
    eng = cabi.Engine()
    l = cabi.lib()
    u = sp.csc_matrix(P.U[0])
    # prolongation before the level count is a state error
    rc = l.gmg_set_prolongation(eng._h, 0, u.shape[0], u.shape[1], cabi._pi(u.indptr), cabi._pi(u.indices), cabi._pd(u.data))
    assert rc == cabi.GMG_ERR_STATE
    assert l.gmg_set_num_levels(eng._h, 1) == 0
    bad = u.indices.copy(); bad[0] = u.shape[0] + 5
    assert l.gmg_set_prolongation(eng._h, 0, u.shape[0], u.shape[1], cabi._pi(u.indptr), cabi._pi(bad), cabi._pd(u.data)) == cabi.GMG_ERR_INVALID
    assert l.gmg_set_prolongation(eng._h, 3, u.shape[0], u.shape[1], cabi._pi(u.indptr), cabi._pi(u.indices), cabi._pd(u.data)) == cabi.GMG_ERR_INVALID
    assert b"" != l.gmg_last_error(eng._h)
    out = C.c_double()
    assert l.gmg_get_timing(eng._h, b"no_such_key", C.byref(out)) == cabi.GMG_ERR_INVALID


# ---------------------------------------------------------------------------------------------- hierarchy builder
@pytest.mark.parametrize("kind", ["torus", "torus-random", "pointcloud"])
def test_hierarchy_properties(cabi, kind):
    """Properties the reference's constructProlongation guarantees (multigrid_solver.cpp:62-469):
    <= 3 parents per row, barycentric weights in [0,1] summing to 1, every accepted level >= lower_bound, chained shapes."""
    from gravo_mg_amd import meshgen
    lb = 40
    if kind == "pointcloud":
        pos = meshgen.torus_points(4000, noise=0.002)
        S, _ = meshgen.knn_graph_laplacian(pos, 8)
    else:
        pos, F = meshgen.torus_mesh(72, 60, order="random" if kind == "torus-random" else "natural")
        S, _ = meshgen.cotan_laplacian(pos, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(pos, neigh, lower_bound=lb)
    assert len(H.U) >= 2
    n = pos.shape[0]
    for k, U in enumerate(H.U):
        assert U.shape[0] == n and lb <= U.shape[1] < n
        n = U.shape[1]
        per_row = np.diff(sp.csr_matrix(U).indptr)
        assert per_row.min() >= 1 and per_row.max() <= 3
        assert U.data.min() >= -1e-12 and U.data.max() <= 1 + 1e-12
        np.testing.assert_allclose(np.asarray(U.sum(axis=1)).ravel(), 1.0, atol=1e-12)
        assert np.all(np.diff(U.indptr) >= 1)                       # every coarse point has children
        assert U.has_sorted_indices
    # coarsening factor of FASTDISK with ratio 8 sits around 6
    assert 3.0 <= pos.shape[0] / H.U[0].shape[1] <= 12.0
    # deterministic
    H2 = cabi.Hierarchy(pos, neigh, lower_bound=lb)
    for a, b in zip(H.U, H2.U):
        assert (a != b).nnz == 0
    t = H.timings()
    assert t["n_vertices"] == pos.shape[0] and t["levels"] == len(H.U) and t["hierarchy"] > 0


def test_hierarchy_options(cabi):
    from gravo_mg_amd import meshgen
    pos, F = meshgen.torus_mesh(48, 40)
    S, _ = meshgen.cotan_laplacian(pos, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    base = cabi.Hierarchy(pos, neigh, lower_bound=40)
    # lower_bound larger than the first sample set -> no levels at all (the reference then has U.size() == 0)
    assert len(cabi.Hierarchy(pos, neigh, lower_bound=1000).U) == 0
    # nested: sample points interpolate themselves with weight 1 (multigrid_solver.cpp:297-300)
    nested = cabi.Hierarchy(pos, neigh, lower_bound=40, nested=True)
    U0 = sp.csr_matrix(nested.U[0])
    assert (np.diff(U0.indptr) == 1).sum() >= U0.shape[1]
    # uniform / inverse-distance weighting keep the sparsity pattern of the barycentric default
    for w in (1, 2):
        alt = cabi.Hierarchy(pos, neigh, lower_bound=40, weighting=w)
        assert alt.U[0].shape == base.U[0].shape
        np.testing.assert_allclose(np.asarray(alt.U[0].sum(axis=1)).ravel(), 1.0, atol=1e-12)
    # UNIFORM: 1/3 (triangle) or 1/2 (edge) everywhere except the "closest three" fallback rows, which use
    # inverse-distance weights under every scheme (multigrid_solver.cpp:415-435)
    uni = cabi.Hierarchy(pos, neigh, lower_bound=40, weighting=1)
    w = np.round(uni.U[0].data, 12)
    assert np.isin(w, [round(1 / 3, 12), 0.5, 1.0]).mean() >= 0.9
    # larger ratio -> larger sampling radius -> fewer coarse points
    assert cabi.Hierarchy(pos, neigh, lower_bound=10, ratio=27.0).U[0].shape[1] < base.U[0].shape[1]
    # out-of-scope samplers are rejected, not silently replaced
    with pytest.raises(cabi.GmgError) as ei:
        cabi.Hierarchy(pos, neigh, lower_bound=40, sampling=1)
    assert ei.value.code == cabi.GMG_ERR_UNSUPPORTED


# ---------------------------------------------------------------------------------------------- Galerkin product
@pytest.mark.parametrize("kind", ["poisson", "bilaplacian", "pointcloud"])
def test_host_galerkin_matches_scipy(cabi, kind):
    P = problems.pointcloud_problem(3000) if kind == "pointcloud" else problems.torus_problem(48, 40, kind, 40)
    A = sp.csc_matrix(P.lhs)
    for U in P.U:
        want = sp.csc_matrix(U.T @ A @ U)
        got = cabi.host_galerkin(A, U)
        assert got.shape == want.shape and got.has_sorted_indices
        assert abs(got - want).max() <= 1e-13 * abs(want).max()
        assert abs(got - got.T).max() <= 1e-13 * abs(want).max()
        # pattern == exact symbolic product (explicit zeros from cancellation are kept, like Eigen's product)
        Uo, Ao = sp.csc_matrix(U, copy=True), sp.csc_matrix(A, copy=True)
        Uo.data[:] = 1.0; Ao.data[:] = 1.0                  # scipy drops exact-zero sums, so count with ones
        assert got.nnz == sp.csc_matrix(Uo.T @ Ao @ Uo).nnz
        A = want


# ---------------------------------------------------------------------------------------------- layout planner
@pytest.mark.parametrize("order", ["natural", "random"])
def test_colour_major_ordering(cabi, order):
    P = problems.torus_problem(48, 40, "poisson", 40, order=order)
    As = [sp.csc_matrix(P.lhs), sp.csc_matrix(P.U[0].T @ P.lhs @ P.U[0])]
    for A in As:
        n = A.shape[0]
        plan = cabi.host_plan_level(A, mode=0)
        new2old, cb = plan["new2old"], plan["color_begin"]
        real = new2old[new2old >= 0]
        assert sorted(real) == list(range(n))                                 # a permutation of the rows
        assert plan["n_pad"] % 64 == 0 and np.all(cb % 64 == 0) and cb[-1] == plan["n_pad"]
        colour = np.empty(n, int)
        for c in range(plan["n_colors"]):
            rows = new2old[cb[c]:cb[c + 1]]
            colour[rows[rows >= 0]] = c
        coo = sp.coo_matrix(A); off = coo.row != coo.col
        assert np.all(colour[coo.row[off]] != colour[coo.col[off]])          # proper colouring
        assert plan["offdiag_nnz"] == A.nnz - n
        assert plan["sell_stored"] >= plan["offdiag_nnz"] and plan["sell_stored"] % 64 == 0
    # valence-6 mesh: 7-point rows, a handful of colours, little SELL padding (4 colours)
    plan = cabi.host_plan_level(As[0], mode=0)
    assert plan["n_colors"] <= 8 and plan["sell_stored"] <= 1.25 * plan["offdiag_nnz"]


@pytest.mark.parametrize("order", ["natural", "random"])
def test_block_ordering_is_compact_and_properly_coloured(cabi, order):
    P = problems.torus_problem(72, 60, "poisson", 40, order=order)
    A = sp.csc_matrix(P.U[0].T @ P.lhs @ P.U[0])
    n = A.shape[0]
    for rows in (64, 128, 256):          # (64: a block is one word per member in the smallest-last colouring; beyond: its bucket form)
        plan = cabi.host_plan_level(A, mode=1, block_rows=rows)
        new2old, bb, rc = plan["new2old"], plan["blk_begin"], plan["row_color"]
        assert sorted(new2old[new2old >= 0]) == list(range(n))
        assert np.all(np.diff(bb) % 64 == 0) and np.all(np.diff(bb) <= rows) and bb[-1] == plan["n_pad"]
        blk_dev = np.repeat(np.arange(plan["n_blocks"]), np.diff(bb))
        real = new2old >= 0
        blk = np.empty(n, int); col = np.empty(n, int)
        blk[new2old[real]] = blk_dev[real]; col[new2old[real]] = rc[real]
        coo = sp.coo_matrix(A); off = coo.row != coo.col
        same = blk[coo.row] == blk[coo.col]
        assert np.all(col[coo.row[same & off]] != col[coo.col[same & off]])      # proper inside every block
        # compact blocks: most couplings stay inside a block even when the input order is random
        assert same[off].mean() >= 0.6, same[off].mean()
        # colours ascend inside a block
        d_blk = np.diff(blk_dev[real]); d_col = np.diff(rc[real].astype(int))
        assert np.all((d_col >= 0) | (d_blk != 0))


def test_smallest_last_in_block_colouring_needs_no_more_colours_and_does_not_depend_on_the_threads(cabi):
    """The block sweeps walk a block's colours one after the other, so the in-block colouring visits a block's rows in smallest-last order (default;
    GMG_BLOCK_COLOURING=bfs: breadth-first, rounds 2-5).  The switches are read once per process: child processes give the breadth-first colouring
    and the one-thread result of the same level."""
    import json, subprocess, sys, os
    code = ("import sys, json, numpy as np, scipy.sparse as sp; sys.path.insert(0, %r); from gravo_mg_amd import cabi; from tests import problems; "
            "P = problems.torus_problem(150, 140, 'poisson', 100); A = sp.csc_matrix(P.U[0].T @ P.lhs @ P.U[0]); out = {}\n"
            "for rows in (64, 256):\n"
            "    plan = cabi.host_plan_level(A, mode=1, block_rows=rows); bb = plan['blk_begin']; rc = plan['row_color']\n"
            "    out[str(rows)] = {'colours': [int(rc[bb[b]:bb[b + 1]].max()) + 1 for b in range(len(bb) - 1)], 'sum': int(np.dot(plan['new2old'].astype(np.int64) + 2, np.arange(len(plan['new2old'])) %% 1000003))}\n"
            "print(json.dumps(out))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(**env):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    sl, bfs, sl1 = run(), run(GMG_BLOCK_COLOURING="bfs"), run(GMG_HOST_THREADS="1")
    for rows in ("64", "256"):
        assert sl[rows] == sl1[rows]                                             # same ordering whatever the number of threads
        a, b = np.array(sl[rows]["colours"]), np.array(bfs[rows]["colours"])
        assert a.shape == b.shape and a.mean() <= b.mean() and a.max() <= b.max()
    assert np.mean(sl["64"]["colours"]) <= 0.95 * np.mean(bfs["64"]["colours"])   # (Galerkin level of a mesh: ~13 % fewer)


def test_missing_diagonal_is_a_numeric_error(cabi):
    A = sp.csc_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 2.0]]))
    A.eliminate_zeros()
    with pytest.raises(cabi.GmgError) as ei:
        cabi.host_plan_level(A, mode=0)
    assert ei.value.code == cabi.GMG_ERR_NUMERIC


# ---------------------------------------------------------------------------------------------- coarsest solver
def test_host_ldlt(cabi):
    P = problems.torus_problem(72, 60, "poisson", 40)
    A = sp.csc_matrix(P.lhs)
    for U in P.U:
        A = sp.csc_matrix(U.T @ A @ U)
    assert 40 <= A.shape[0] <= 400
    rng = np.random.default_rng(0)
    b = rng.standard_normal((A.shape[0], 3))
    x, nnzL = cabi.host_ldlt_solve(A, b)
    assert np.linalg.norm(A @ x - b) <= 1e-11 * (spla.norm(A) * np.linalg.norm(x))
    assert 0 < nnzL < A.shape[0] * (A.shape[0] - 1) // 2
    # a well-conditioned SPD system to full accuracy, and 1-D right-hand sides keep their shape
    B = sp.csc_matrix(problems.torus_problem(40, 36, "smoothing", 60).lhs)
    y = rng.standard_normal(B.shape[0])
    z, nnzB = cabi.host_ldlt_solve(B, y)
    assert z.shape == y.shape and np.linalg.norm(B @ z - y) <= 1e-12 * np.linalg.norm(y)
    assert nnzB < 40 * B.shape[0]                       # fill-reducing ordering: far from dense
    # singular matrix -> loud numeric error, no garbage
    Z = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 1.0]]))
    with pytest.raises(cabi.GmgError) as ei:
        cabi.host_ldlt_solve(Z, np.ones(2))
    assert ei.value.code == cabi.GMG_ERR_NUMERIC


def test_host_ldlt_nested_dissection_path(cabi):
    """Above 2500 unknowns the coarsest solver orders by nested dissection (BFS level-set separators)."""
    P = problems.torus_problem(120, 100, "smoothing", 2000)
    A = sp.csc_matrix(P.lhs)
    assert A.shape[0] == 12000
    rng = np.random.default_rng(1)
    b = rng.standard_normal((A.shape[0], 2))
    x, nnzL = cabi.host_ldlt_solve(A, b)
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    assert nnzL < 60 * A.shape[0]                     # O(n log n) fill on a 2-D mesh graph, far from the dense n^2/2
    # block-diagonal (disconnected) input exercises the component peeling
    B = sp.block_diag([A[:3000, :3000] + sp.identity(3000), sp.identity(50) * 2.0, A[:2600, :2600] + sp.identity(2600)], format="csc")
    y = rng.standard_normal(B.shape[0])
    z, _ = cabi.host_ldlt_solve(B, y)
    assert np.linalg.norm(B @ z - y) <= 1e-12 * np.linalg.norm(y)


def test_host_ldlt_many_right_hand_sides_and_galerkin_operator(cabi):
    """The supernodal solver on what it is built for (a Galerkin coarsest operator, ~20 entries per row) with six
    right-hand sides (chunks of four): every column equals its own single-column solve and scipy's."""
    P = problems.torus_problem(140, 120, "smoothing", 1500)
    A = sp.csc_matrix(P.lhs)
    for U in P.U:
        A = sp.csc_matrix(U.T @ A @ U)
    assert 1500 <= A.shape[0] <= 12000 and A.nnz / A.shape[0] > 12
    rng = np.random.default_rng(4)
    B = rng.standard_normal((A.shape[0], 6))
    X, nnzL = cabi.host_ldlt_solve(A, B)
    lu = spla.splu(A)
    for c in range(6):
        x1, _ = cabi.host_ldlt_solve(A, B[:, c].copy())
        assert np.array_equal(X[:, c], x1)
        assert np.linalg.norm(X[:, c] - lu.solve(B[:, c])) <= 1e-10 * np.linalg.norm(X[:, c])
    assert nnzL > A.nnz // 2


def test_host_ldlt_team_back_substitution_gives_the_same_bits():
    """The coarsest back-substitution runs the parts of the elimination tree (2 .. 8 sets of disjoint subtrees, by the size of the
    factor) on a team of spinning threads inside a solve (SpinTeam); the arithmetic must not depend on how many threads share the
    parts.  The library's probe (gmg_host_ldlt_probe, gravomg_hip_internal.h) solves with 1, 2, 3, ... threads and reports the largest difference."""
    import re
    from gravo_mg_amd import cabi
    m = 90
    T = sp.diags([-1.0, 2.3, -1.0], [-1, 0, 1], shape=(m, m))
    A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
    b = np.random.default_rng(3).standard_normal(m * m)
    x, nnz = cabi.host_ldlt_solve(A, b)
    report = cabi.host_ldlt_probe(A, b, reps=3)

    class out:                      # (the report used to arrive on a subprocess's stderr)
        stderr = report
        stdout = "residual %r" % float(np.linalg.norm(A @ x - b) / np.linalg.norm(b))
    m1 = re.search(r"(\d+) parts of the elimination tree, panel entries in the lightest / heaviest part / above them: (\d+) / (\d+) / (\d+)", out.stderr)
    assert m1, out.stderr[-2000:]
    parts, lo, hi, top = (int(m1.group(k)) for k in (1, 2, 3, 4))
    assert parts >= 3 and lo > 0 and lo >= 0.5 * hi, (parts, lo, hi, top)          # a real split, reasonably balanced
    diffs = re.findall(r"(\d+) threads: .* max \|difference\| to the one-thread solve ([0-9.e+-]+)", out.stderr)
    assert len(diffs) >= 2 and all(float(d) == 0.0 for _, d in diffs), out.stderr[-2000:]
    assert float(re.search(r"residual ([0-9.e+-]+)", out.stdout).group(1)) <= 1e-10


def test_host_ldlt_threaded_factorisation_gives_the_same_bits():
    """The numeric factorisation runs the parts of the elimination tree side by side and shares the update phase of every supernode of
    the top between a team of threads (host_ldlt.hpp::numeric): the updates of a supernode are subtracted in ascending order of its
    descendants whoever computes them, so the factor -- here seen through a solve -- has the same bits for 1, 3 and 8 threads, and is
    the factor of the matrix (residual against the input, solution against scipy's LU)."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, sys\n"
        "from gravo_mg_amd import cabi\n"
        "from tests import problems\n"
        "P = problems.torus_problem(96, 80, 'smoothing', 30)\n"
        "A = sp.csc_matrix(P.lhs)\n"
        "A = sp.csc_matrix(P.U[0].T @ A @ P.U[0])\n"          # a Galerkin operator (M + tau S: well conditioned): ~20 entries per row, 1 250 unknowns
        "m = 70\n"
        "T = sp.diags([-1.0, 2.2, -1.0], [-1, 0, 1], shape=(m, m))\n"
        "G = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()\n"      # and a 4 900-unknown grid: several parts, a top of separators
        "for M in (A, G):\n"
        "    b = np.random.default_rng(3).standard_normal((M.shape[0], 2))\n"
        "    x, nnz = cabi.host_ldlt_solve(M, b)\n"
        "    assert np.linalg.norm(M @ x - b) <= 1e-10 * np.linalg.norm(b)\n"
        "    assert np.linalg.norm(x - spla.splu(M).solve(b)) <= 1e-9 * np.linalg.norm(x)\n"
        "    sys.stdout.write(x.tobytes().hex() + '\\n')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for threads in ("1", "3", "8"):
        env = dict(os.environ, GMG_LDLT_THREADS=threads)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out.stdout)
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 1000


def test_fine_block_rule_on_the_host(cabi):
    """The rule behind gmg_config::block_fine (level 0 blocked too): kNN graph Laplacians qualify -- long rows, positive diagonal, no positive
    off-diagonal entry --; a triangle-mesh operator has too few entries per row, a Bilaplacian has positive second-ring entries, and one
    positive coupling or a non-positive diagonal entry disqualifies a kNN operator (the block sweep is then no regular splitting)."""
    import scipy.sparse as sp
    from gravo_mg_amd import meshgen
    P = problems.pointcloud_problem(3000)
    assert cabi.host_fine_block_rule(P.lhs) == (True, 0)
    T = problems.torus_problem(48, 40, "poisson", 60)
    assert cabi.host_fine_block_rule(T.lhs) == (False, 1)
    B = problems.torus_problem(48, 40, "bilaplacian", 40)
    assert cabi.host_fine_block_rule(B.lhs) == (False, 2)
    A = sp.lil_matrix(P.lhs)
    nb = sp.csc_matrix(P.lhs)[:, 5].indices; j = int(nb[nb != 5][0])
    A[5, j] = A[j, 5] = 1e-6
    assert cabi.host_fine_block_rule(sp.csc_matrix(A)) == (False, 2)
    A = sp.lil_matrix(P.lhs); A[7, 7] = 0.0
    assert cabi.host_fine_block_rule(sp.csc_matrix(A)) == (False, 2)
    S, mass = meshgen.knn_graph_laplacian(meshgen.torus_points(2000, noise=0.002), 12)      # smoothing system of a denser graph
    assert cabi.host_fine_block_rule(meshgen.smoothing_system(S, mass, np.zeros((2000, 3)))[0]) == (True, 0)



@pytest.mark.parametrize("kind", ["grid", "coarsest"])
def test_device_factor_layout_and_schedule_on_the_host(kind):
    """The dense inverse of the coarsest operator is built on the GPU from the host's sparse factor in a layout of its own (chunks of <= 8
    columns, row-major values, levels of the elimination tree: SupernodalLDLT::export_device_factor) by a kernel that carries the identity
    through it (setup_kernels.hip.hpp::coarse_inverse_tiles).  The same algorithm on the same arrays, on the host, one column at a time
    (emulate_device_column), against the back-substitution of the unit vectors -- the layout and the schedule without a GPU; the GPU suite has
    the kernel itself (tests/test_gpu_parity.py::test_device_built_coarse_inverse_against_the_host_factor)."""
    import re
    from gravo_mg_amd import cabi
    if kind == "grid":
        m = 70
        T = sp.diags([-1.0, 2.3, -1.0], [-1, 0, 1], shape=(m, m))
        A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
    else:                               # a real coarsest Galerkin operator (~21 entries per row, nnz(L) / n ~ 100)
        P = problems.torus_problem(130, 120, "smoothing", 300)
        A = P.lhs
        for U in P.U:
            A = cabi.host_galerkin(A, U)
        A = sp.csc_matrix(A)
    b = np.random.default_rng(5).standard_normal(A.shape[0])
    report = cabi.host_ldlt_probe(A, b, reps=1)
    m1 = re.search(r"device factor layout: (\d+) chunks of <= 8 columns in (\d+) levels; (\d+) columns .* max \|difference\| ([0-9.e+-]+) of max \|entry\| ([0-9.e+-]+)", report)
    assert m1, report[-1500:]
    chunks, levels, cols, diff, scale = int(m1.group(1)), int(m1.group(2)), int(m1.group(3)), float(m1.group(4)), float(m1.group(5))
    assert chunks >= A.shape[0] // 8 and 2 <= levels <= chunks and cols == 24
    assert diff <= 1e-11 * scale, (diff, scale)


def test_colouring_ahead_of_the_inspection_gives_the_same_ordering_and_survives_bad_arrays(cabi):
    """A cold gmg_set_system starts the greedy colouring of level 0 at entry, beside the inspection of the caller's arrays (gmg_config::color_ahead,
    host_plan.hpp::greedy_coloring_ahead): the same first-fit colours as the loop that runs after the inspection, hence the same ordering; every
    pointer and index is range-checked, so arrays the inspection would reject end the loop instead of taking it out of bounds; beyond 64 colours it
    gives up and the general loop colours."""
    import scipy.sparse as sp
    from gravo_mg_amd import meshgen
    V, F = meshgen.torus_mesh(150, 140)
    S, _ = meshgen.cotan_laplacian(V, F)
    rng = np.random.default_rng(3)
    R = sp.random(4000, 4000, density=0.002, random_state=5, format="csr")
    R = (R + R.T + sp.identity(4000)).tocsc()
    for A in (S.tocsc(), R):
        A.sort_indices()
        n = A.shape[0]
        ref = cabi.host_plan_level(A, mode=3)
        got = cabi.host_plan_ahead(n, A.indptr, A.indices)
        assert got["colored_ahead"] and got["n_colors"] == ref["n_colors"] and got["n_pad"] == ref["n_pad"]
        assert np.array_equal(got["color_begin"], ref["color_begin"]) and np.array_equal(got["new2old"], ref["new2old"])
    # a complete graph on 70 vertices needs 70 colours: the 64-bit mask runs out, the general loop takes over -- same ordering
    K = sp.csc_matrix(np.ones((70, 70)))
    K.sort_indices()
    ref = cabi.host_plan_level(K, mode=3)
    got = cabi.host_plan_ahead(70, K.indptr, K.indices)
    assert not got["colored_ahead"] and got["n_colors"] == ref["n_colors"] == 70 and np.array_equal(got["new2old"], ref["new2old"])
    # arrays that fail the inspection: an index out of range, a negative one, pointers that go backwards, pointers beyond the claimed entry count
    A = S.tocsc(); A.sort_indices()
    n = A.shape[0]
    for spoil in ("index-high", "index-negative", "pointer-backwards", "pointer-beyond"):
        ptr, idx = A.indptr.astype(np.int32).copy(), A.indices.astype(np.int32).copy()
        if spoil == "index-high":
            idx[ptr[n // 2]] = n + 5          # (first entry of a row: read before the `j >= i` stop)
        elif spoil == "index-negative":
            idx[ptr[n // 3]] = -7
        elif spoil == "pointer-backwards":
            ptr[n // 2] = ptr[n // 2 - 1] - 3
        else:
            ptr[n // 2 + 1] = ptr[n] + 1000
            ptr[n // 2 + 2:] = np.maximum(ptr[n // 2 + 2:], ptr[n // 2 + 1])
            ptr[n] = A.indptr[n]
        with pytest.raises(cabi.GmgError):
            cabi.host_plan_ahead(n, ptr, idx)
