"""Level 0 as a blocked level chosen by the engine (gmg_config::block_fine, round 4): an operator with long rows whose signs make
the block-hybrid sweep a regular splitting (kNN graph Laplacians of point clouds: BASELINE config 3) runs ONE launch per sweep on
level 0 instead of one per colour (11 at 2 M points).  The smoother is still a parallel re-ordering of the reference's Gauss-Seidel
(multigrid_solver.cpp:1194-1226): Gauss-Seidel inside a 64-row block, Jacobi between blocks -- the iteration the Galerkin levels
have run since round 1 --, so parity is stated the same way: sweep == its matrix form with the oracle's residual, V-cycle == the
model assembled from the oracle's operators cycle by cycle, solve == the reference algorithm's solution to the tolerance.

Tolerances: matrix form 1e-12 (relative, fp64 sums of <= 20 terms); per cycle the backward-error form
||A (x_gpu - x_model)|| <= 1e-12 ||A|| ||x|| of the Poisson systems (tau M + S, tau = 1e-6: ||x|| / ||b|| ~ 1e8) and 1e-6 forward."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from tests import problems

pytestmark = pytest.mark.gpu


def _cloud(n=6000, lower_bound=100):
    return problems.pointcloud_problem(n, lower_bound=lower_bound)


def _engine(cabi, P, **kw):
    e = cabi.Engine(**kw)
    e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
    return e


def test_the_rule_picks_point_clouds_only(cabi):
    """Long rows AND Stieltjes signs: the kNN Laplacian qualifies; a triangle mesh (7 entries per row), a Bilaplacian (19 per row,
    positive second-ring entries) and a kNN operator with one positive off-diagonal entry do not; block_fine = 0 switches the rule off."""
    P = _cloud()
    assert P.lhs.nnz >= 9 * P.lhs.shape[0]
    e = _engine(cabi, P)
    assert e.timing("fine_level_blocked") == 1.0 and e.level_blocks(0) is not None
    bb, _ = e.level_blocks(0)
    assert np.all(np.diff(bb) == 64)                                   # every block full (runs of 64 points of the cluster order) but possibly the last
    off = _engine(cabi, P, block_fine=0)
    assert off.timing("fine_level_blocked") == 0.0 and off.level_blocks(0) is None and off.level_info(0)["n_colors"] >= 8
    T = problems.torus_problem(96, 80, "poisson", 30)
    assert _engine(cabi, T).level_blocks(0) is None
    B = problems.torus_problem(64, 60, "bilaplacian", 40)
    assert B.lhs.nnz >= 9 * B.lhs.shape[0] and _engine(cabi, B).level_blocks(0) is None
    # one positive coupling (symmetric): the sign test fails, the colour-major sweep stays
    A = sp.lil_matrix(P.lhs)
    i = 17; nb = sp.csc_matrix(P.lhs)[:, i].indices; j = int(nb[nb != i][0])
    A[i, j] = A[j, i] = 1e-3 * abs(P.lhs[i, i])
    bad = cabi.Engine(); bad.set_prolongations(P.U); bad.set_mass(P.mass); bad.set_system(sp.csc_matrix(A))
    assert bad.level_blocks(0) is None and bad.timing("fine_level_blocked") == 0.0


def test_the_chosen_layout_is_the_one_block_from_level_0_builds(cabi):
    """The rule only decides: orderings, blocks and iterates are those of an engine told to block every level."""
    P = _cloud()
    a, b = _engine(cabi, P), _engine(cabi, P, block_from_level=0)
    for k in range(len(P.U)):
        (na, ca), (nb, cb) = a.level_ordering(k), b.level_ordering(k)
        assert np.array_equal(na, nb) and np.array_equal(ca, cb)
        assert np.array_equal(a.level_blocks(k)[0], b.level_blocks(k)[0]) and np.array_equal(a.level_blocks(k)[1], b.level_blocks(k)[1])
    xa, ita, ra, _ = a.solve(P.rhs, tol=1e-4)
    xb, itb, rb, _ = b.solve(P.rhs, tol=1e-4)
    assert ita == itb and ra == rb and np.array_equal(xa, xb)


def test_blocks_are_compact_patches(cabi):
    """Runs of the cluster order are patches of the surface, not strips or scattered points: most couplings of a block's rows stay
    inside the block (a contiguous run of a random numbering keeps ~64 / n of them)."""
    P = _cloud(20_000, lower_bound=200)
    e = _engine(cabi, P)
    n2o, _ = e.level_ordering(0)
    bb, _ = e.level_blocks(0)
    blk_of_new = np.repeat(np.arange(len(bb) - 1), np.diff(bb))
    blk = np.full(P.lhs.shape[0], -1); real = n2o >= 0
    blk[n2o[real]] = blk_of_new[real]
    assert np.all(blk >= 0)
    C = sp.coo_matrix(P.lhs)
    offd = C.row != C.col
    inside = float(np.mean(blk[C.row[offd]] == blk[C.col[offd]]))
    print("in-block share of the off-diagonal entries:", inside)
    assert inside >= 0.45


@pytest.mark.parametrize("d", [1, 3])
def test_fine_level_sweep_matches_matrix_form(cabi, oracle, d):
    """One sweep of the blocked level 0 == x + T^-1 (b - A x), T = D + strict lower triangle of A restricted to the block diagonal in
    device order (residual from the oracle, T^-1 from scipy); also from the zero iterate and for two sweeps."""
    P = _cloud()
    e = _engine(cabi, P)
    A = sp.csr_matrix(P.lhs)
    n2o, _ = e.level_ordering(0)
    bb, _ = e.level_blocks(0)
    real = n2o >= 0
    blk = np.repeat(np.arange(len(bb) - 1), np.diff(bb))[real]
    order = n2o[real]
    Ap = A[order][:, order].tocoo()
    keep = (blk[Ap.row] == blk[Ap.col]) & (Ap.col <= Ap.row)
    data = Ap.data[keep].copy()
    T = sp.csr_matrix((data, (Ap.row[keep], Ap.col[keep])), shape=Ap.shape)
    rng = np.random.default_rng(5)
    b = rng.standard_normal((A.shape[0], d)); x0 = rng.standard_normal((A.shape[0], d))
    for start in (x0, np.zeros_like(x0)):
        x = start.copy()
        for sweeps in (1, 2):
            r = oracle.residual(P.lhs, b, x)
            step = np.empty_like(x); step[order] = spla.spsolve_triangular(T, r[order], lower=True)
            x = x + step
            got = e.smooth(0, b, start, sweeps)
            assert np.linalg.norm(got - x) <= 1e-12 * np.linalg.norm(x), (sweeps, np.linalg.norm(got - x) / np.linalg.norm(x))


def test_cycles_match_the_model_and_the_solve_the_reference_algorithm(cabi, oracle):
    from tests.vcycle_model import VcycleModel
    P = _cloud(12_000, lower_bound=150)
    e = _engine(cabi, P)
    assert e.level_blocks(0) is not None and e.num_levels >= 2
    M = VcycleModel(e, P.U, P.mass, P.lhs, oracle, e.gs_omega)
    nA = spla.norm(P.lhs)
    x = P.rhs.copy()
    for cyc in range(3):
        xg, xm = e.vcycle(P.rhs, x), M.vcycle(P.rhs, x)
        assert np.linalg.norm(P.lhs @ (xg - xm)) <= 1e-12 * nA * np.linalg.norm(xm), cyc
        assert np.linalg.norm(xg - xm) <= 1e-6 * np.linalg.norm(xm), cyc
        assert abs(e.residual_norm(P.rhs, xg, 2) - oracle.residual_check(P.lhs, P.mass, P.rhs, xg, 2)) <= 1e-7
        x = xm
    xs, it, res, conv = e.solve(P.rhs, tol=1e-4)
    O = oracle.Hierarchy(P.U, P.mass); O.set_system(P.lhs)
    xo, ito, reso, _ = O.solve(P.rhs, tol=1e-4)
    chk = oracle.residual_check(P.lhs, P.mass, P.rhs, xs, 2)
    assert res <= 1e-4 and abs(chk - res) <= 1e-3 * res + 1e-9
    assert it <= ito + 2 and np.all(np.diff(conv[:, 1]) < 0)
    m = P.mass[:, None]
    assert np.sqrt((m * (xs - xo) ** 2).sum() / (m * xo ** 2).sum()) <= 20 * 1e-4
    # not slower to converge than the colour-major engine by more than two cycles, and the same answer
    x2, it2, res2, _ = _engine(cabi, P, block_fine=0).solve(P.rhs, tol=1e-4)
    assert it <= it2 + 2 and np.sqrt((m * (xs - x2) ** 2).sum() / (m * x2 ** 2).sum()) <= 20 * 1e-4


def test_same_pattern_systems_keep_or_drop_the_blocks_with_their_signs(cabi, oracle):
    """A system with the live sparsity pattern refreshes values only -- unless level 0 was blocked by the rule and the new values fail its
    sign test: then the whole layout is rebuilt colour-major (and a colour-major layout is kept for later systems of that pattern: the
    rule is evaluated when a layout is built).  Either way the solve is the new system's."""
    P = _cloud()
    e = _engine(cabi, P)
    assert e.level_blocks(0) is not None
    A2 = sp.csc_matrix(P.lhs); A2.data = A2.data * 1.5
    e.set_system(A2)
    assert e.timing("setup_values_only") == 1.0 and e.level_blocks(0) is not None
    x, it, res, _ = e.solve(P.rhs, tol=1e-4)
    assert res <= 1e-4 and abs(oracle.residual_check(A2, P.mass, P.rhs, x, 2) - res) <= 1e-3 * res + 1e-9
    A3 = sp.csc_matrix(P.lhs)
    i = 29; nb = A3[:, i].indices; j = int(nb[nb != i][0])
    A3 = sp.lil_matrix(A3); A3[i, j] = A3[j, i] = 1e-3 * abs(P.lhs[i, i]); A3 = sp.csc_matrix(A3)
    assert A3.nnz == P.lhs.nnz
    e.set_system(A3)
    assert e.timing("setup_values_only") == 0.0 and e.level_blocks(0) is None and e.timing("fine_level_blocked") == 0.0
    x, it, res, _ = e.solve(P.rhs, tol=1e-4)
    assert res <= 1e-4 and abs(oracle.residual_check(A3, P.mass, P.rhs, x, 2) - res) <= 1e-3 * res + 1e-9
    e.set_system(P.lhs)                                                 # and back: same pattern as the live (colour-major) layout -- values only, the layout stays
    assert e.timing("setup_values_only") == 1.0 and e.level_blocks(0) is None
    x, it, res, _ = e.solve(P.rhs, tol=1e-4)
    assert res <= 1e-4 and abs(oracle.residual_check(P.lhs, P.mass, P.rhs, x, 2) - res) <= 1e-3 * res + 1e-9


@pytest.mark.parametrize("kw", [dict(inner_precision=1), dict(use_graph=True), dict(coarse_mode=1), dict(device_setup=False)],
                         ids=["fp32-inner-cycle", "hipgraph", "device-coarse-apply", "host-planner"])
def test_engine_variants_on_a_blocked_fine_level(cabi, oracle, kw):
    """The other ways of running the cycle meet a blocked level 0 too: the fp32 inner cycle of the mixed-precision iteration (fp32 twins of
    the block storage), legs replayed from hipGraphs, the coarsest solve applied on the device, layouts from the host planner -- the same
    answer as the plain engine (bitwise where the arithmetic is the same), confirmed by the oracle's residual check."""
    P = _cloud()
    ref = _engine(cabi, P)
    e = _engine(cabi, P, **kw)
    assert e.level_blocks(0) is not None and np.array_equal(e.level_ordering(0)[0], ref.level_ordering(0)[0])
    x, it, res, _ = e.solve(P.rhs, tol=1e-4)
    xr, itr, resr, _ = ref.solve(P.rhs, tol=1e-4)
    assert res <= 1e-4 and abs(oracle.residual_check(P.lhs, P.mass, P.rhs, x, 2) - res) <= 1e-3 * res + 1e-9
    m = P.mass[:, None]
    if "inner_precision" in kw:
        # the fp32 inner cycle costs this system (tau M + S, tau = 1e-6: ||x|| / ||b|| ~ 1e8, far from what mixed precision is for) cycles in either
        # layout, more in the blocked one: 12 against 9 in fp64 here, 8 against 7 colour-major
        _, itc, resc, _ = _engine(cabi, P, block_fine=0, **kw).solve(P.rhs, tol=1e-4)
        print("fp32 inner cycle: blocked", it, "colour-major", itc, "fp64 blocked", itr)
        assert resc <= 1e-4 and it <= itr + 4 and np.sqrt((m * (x - xr) ** 2).sum() / (m * xr ** 2).sum()) <= 20 * 1e-4
    elif "coarse_mode" in kw:
        assert abs(it - itr) <= 1 and np.sqrt((m * (x - xr) ** 2).sum() / (m * xr ** 2).sum()) <= 20 * 1e-4
    else:
        assert it == itr and res == resr and np.array_equal(x, xr)
    x3, it3, res3, _ = e.solve(np.repeat(P.rhs, 3, axis=1) * np.array([[1.0, -2.0, 0.5]]), tol=1e-4)      # d = 3 through the same layout
    assert res3 <= 1e-4 and np.allclose(x3[:, :1], x, rtol=0, atol=1e-6 * np.abs(x).max())


def test_the_distributed_path_takes_a_blocked_level_0(cabi):
    """A blocked level 0 is partitioned by runs of whole 64-row blocks of the entry-parallel sweep (round 6: the same smoother at every rank count,
    tests/test_gpu_p2p.py runs it): the block count is padded to a multiple of the rank count (gmg_config::row_align = 64 P); a colour sweep
    is refused on such a level (its sweep is a block sweep)."""
    P = _cloud()
    e = _engine(cabi, P, row_align=128)
    e.dist_setup(0, 2)
    assert e.level_blocks(0) is not None and (e.level_info(0)["n_pad"] // 64) % 2 == 0
    three = _engine(cabi, P, row_align=192)
    three.dist_setup(1, 3)
    assert (three.level_info(0)["n_pad"] // 64) % 3 == 0
    _engine(cabi, P, row_align=128, block_fine=0).dist_setup(0, 2)


def test_both_in_block_colouring_orders_solve_the_same_systems(cabi):
    """The in-block colouring of the block sweeps follows a smallest-last order (default) or the breadth-first order of rounds 2-5
    (GMG_BLOCK_COLOURING=bfs; the switch is read once per process, hence the child): two orderings of the same block Gauss-Seidel smoother -- both
    converge to the same solution in (about) the same number of cycles, and the default needs no more colours on any level."""
    import json, os, subprocess, sys
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from gravo_mg_amd import cabi; from tests import problems; out = {}\n"
            "for name, P in (('torus', problems.torus_problem(300, 280, 'poisson', 100)), ('cloud', problems.pointcloud_problem(9000, 8, 120))):\n"
            "    e = cabi.Engine(); e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)\n"
            "    x, it, res, conv = e.solve(P.rhs, tol=1e-6, max_iter=100)\n"
            "    out[name] = {'it': int(it), 'res': float(res), 'x': np.asarray(x).ravel()[::97].tolist(), 'colours': [int(e.level_info(k).get('n_colors') or 0) for k in range(e.num_levels + 1)]}\n"
            "print(json.dumps(out))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(**env):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    sl, bfs = run(), run(GMG_BLOCK_COLOURING="bfs")
    for name in ("torus", "cloud"):
        a, b = sl[name], bfs[name]
        assert a["res"] <= 1e-6 and b["res"] <= 1e-6 and abs(a["it"] - b["it"]) <= 1, (name, a["it"], b["it"])
        xa, xb = np.array(a["x"]), np.array(b["x"])
        assert np.linalg.norm(xa - xb) <= 1e-4 * np.linalg.norm(xb), name
        assert all(ca <= cb for ca, cb in zip(a["colours"][1:], b["colours"][1:])), (name, a["colours"], b["colours"])
