"""BASELINE.json's other configs at FULL size (synthetic stand-ins, gravo_mg_amd.meshgen.baseline_config): config 1 literally
(the demos' smoothing call, 36 100 vertices, n x 3 right-hand side), the 720 k mesh Poisson problem, the 2 M point cloud, the 3 M mesh in random vertex order and the 3 M Bilaplacian (fp64 and mixed precision,
with the reference's tau = 1e-3 and with tau = 1e-9, for which the reference iteration contracts).

The 1-core oracle finishes a full solve of these in seconds (0.2 s per V-cycle at 3 M vertices), so besides the
size-independent properties the test runs the REFERENCE ALGORITHM itself on the same input:
  * both reach the reference's stopping test (M-norm of the residual <= 1e-4, multigrid_solver.cpp:1408-1419); the V-cycle
    counts are recorded side by side -- the device's smoother is a parallel re-ordering of the reference's Gauss-Seidel, so
    the counts may differ; asserted: device <= oracle + 1;
  * the two solutions agree to 20 x tol in the M-norm;
  * the oracle's one-pass residualCheck of the device solution equals the device's own residue (all four norm types);
  * the residual history contracts monotonically; the V-cycle is affine in (b, x).
Per-cycle agreement to rounding (the check that also works where the iteration does not contract) is in
tests/test_gpu_cycle_model.py at 109 k vertices.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RECORD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fullsize_configs.jsonl")


def _record(**kw):
    try:
        os.makedirs(os.path.dirname(RECORD), exist_ok=True)
        with open(RECORD, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module", params=["1", "2", "3", "4r", "5b", "5"])
def case(request, cabi):
    from gravo_mg_amd import meshgen
    name, pos, S, mass, lhs, rhs = meshgen.baseline_config(request.param)
    H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
    eng = cabi.Engine()
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    yield dict(cfg=request.param, name=name, mass=mass, H=H, lhs=lhs, rhs=rhs, eng=eng)
    eng.close()


def _mdist(mass, x, y):
    m = mass[:, None]
    return float(np.sqrt((m * (x - y) ** 2).sum() / (m * y ** 2).sum()))


def test_solve_against_the_reference_algorithm(case, cabi, oracle):
    eng, lhs, rhs, mass, cfg = case["eng"], case["lhs"], case["rhs"], case["mass"], case["cfg"]
    O = oracle.Hierarchy(case["H"].U, mass)
    O.set_system(lhs)
    if cfg == "5":
        # The reference's own parameters (tau = 1e-3): the reference iteration does not reach 1e-4 from x0 = rhs at this size
        # (the residue starts at ~1e3 and, over the first cycles, GROWS -- on the oracle and on the device alike).
        # Parity is "same behaviour": same order of magnitude cycle by cycle, same trend, residues that the oracle
        # confirms; fp32 inner cycles follow the fp64 history.  (Per-cycle agreement to rounding: test_gpu_cycle_model.py.)
        eng.load_problem(rhs, rhs)
        hist = eng.run_cycles(8, 2)
        x = eng.fetch_solution()
        chk = oracle.residual_check(lhs, mass, rhs, x, 2)
        assert abs(chk - hist[-1]) <= 1e-6 * chk
        xo, ito, reso, convo = O.solve(rhs, tol=0.0, max_iter=8)
        ratio = hist / convo[:, 1]
        assert np.all((ratio > 0.4) & (ratio < 2.5))
        assert np.sign(hist[-1] - hist[2]) == np.sign(convo[-1, 1] - convo[2, 1])     # same trend
        mix = cabi.Engine(inner_precision=1)
        mix.use_hierarchy(case["H"]); mix.set_mass(mass); mix.set_system(lhs)
        mix.load_problem(rhs, rhs)
        hmix = mix.run_cycles(8, 2)
        np.testing.assert_allclose(hmix, hist, rtol=5e-3)
        assert abs(oracle.residual_check(lhs, mass, rhs, mix.fetch_solution(), 2) - hmix[-1]) <= 1e-6 * hmix[-1]
        mix.close()
        _record(config=case["name"], gpu_history=[float(v) for v in hist], mixed_history=[float(v) for v in hmix],
                oracle_history=[float(v) for v in convo[:, 1]])
        return
    # tau = 1e-9 Bilaplacian: the reference iteration contracts, but by ~3 % per cycle at this size -- neither side reaches 1e-4
    # within the reference's max_iter = 100 (oracle 4.0e-3, device 2.9e-3 after 100 cycles); the convergent leg stops at 3e-2
    tol = 3e-2 if cfg == "5b" else 1e-4
    x, it, res, conv = eng.solve(rhs, tol=tol, stop_type=2, max_iter=100)
    xo, ito, reso, convo = O.solve(rhs, tol=tol, stop_type=2, max_iter=100)
    _record(config=case["name"], gpu_iterations=int(it), oracle_iterations=int(ito), gpu_history=[float(v) for v in conv[:, 1]],
            oracle_history=[float(v) for v in convo[:, 1]], solution_distance_M=_mdist(mass, x, xo))
    assert res <= tol and reso <= tol and it < 100
    assert it <= ito + 1, (it, ito)
    assert np.all(np.diff(conv[:, 1]) < 0)
    assert _mdist(mass, x, xo) <= 20 * tol
    for t in (0, 1, 2, 3):
        want = oracle.residual_check(lhs, mass, rhs, x, t)
        assert abs(eng.residual_norm(rhs, x, t) - want) <= 1e-3 * want + 1e-7
    if cfg == "5b":
        # config 5 proper: the fp32 inner V-cycle inside the fp64 defect-correction loop reaches the same stopping test
        mix = cabi.Engine(inner_precision=1)
        mix.use_hierarchy(case["H"]); mix.set_mass(mass); mix.set_system(lhs)
        xm, itm, resm, convm = mix.solve(rhs, tol=tol, stop_type=2, max_iter=100)
        assert resm <= tol and abs(itm - it) <= 1
        assert abs(oracle.residual_check(lhs, mass, rhs, xm, 2) - resm) <= 1e-3 * resm + 1e-7
        assert _mdist(mass, xm, xo) <= 20 * tol
        _record(config=case["name"] + " (mixed precision)", gpu_iterations=int(itm), oracle_iterations=int(ito),
                gpu_history=[float(v) for v in convm[:, 1]])
        mix.close()


def test_one_cycle_matches_the_model_at_full_size(case, oracle):
    """Per-cycle parity at full size for BASELINE config 1 (the demos' call, d = 3), config 4 in random vertex order and config 5
    with the reference's tau = 1e-3 (where the iteration does not contract, so solve-level checks say little): one V-cycle of
    the default engine from x0 = rhs equals the model of the same iteration assembled from the oracle's operators and the
    device's orderings (tests/vcycle_model.py).  Bounds: the backward error ||A (x_gpu - x_model)|| <= 1e-12 ||A|| ||x|| of
    tests/test_gpu_cycle_model.py for every system; forward, what the conditioning leaves of it at 3 M vertices -- smoothing
    1e-10, Poisson (cond ~ 1e6 x the mesh's) 1e-5 (measured 4.9e-6; 1e-6 holds in natural order), Bilaplacian (cond ~ n^2:
    1.2e-10 at 109 k vertices, x 760 at 3 M) 1e-7 (measured 2.4e-8).  (Why constants and not kappa x eps: the only cheap rigorous
    estimate, lambda_max(Gershgorin) / (tau min M_ii), is ~3e13 for the Poisson system at 3 M vertices -- times the backward error 1e-12 that
    "bounds" the forward error by 30; the measured 5e-6 reflects that the rounding errors of a cycle have almost no component along the
    few near-null vectors.  The backward-error assertion is the check; the forward constants only keep it from rotting.)"""
    import scipy.sparse.linalg as spla
    from tests.vcycle_model import VcycleModel
    cfg = case["cfg"]
    if cfg not in ("1", "4r", "5"):
        pytest.skip("covered by the solve against the reference algorithm")
    eng, lhs, rhs, mass = case["eng"], case["lhs"], case["rhs"], case["mass"]
    M = VcycleModel(eng, case["H"].U, mass, lhs, oracle, eng.gs_omega)
    xg = eng.vcycle(rhs, rhs)
    xm = M.vcycle(rhs, rhs.copy())
    d = np.linalg.norm(xg - xm) / np.linalg.norm(xm)
    _record(config=case["name"], one_cycle_model_distance=float(d))
    back = float(np.linalg.norm(lhs @ (xg - xm)) / (spla.norm(lhs) * np.linalg.norm(xm)))
    _record(config=case["name"], one_cycle_model_backward_error=back)
    assert back <= 1e-12, (back, d)
    assert d <= {"1": 1e-10, "4r": 1e-5, "5": 1e-7}[cfg], d
    want = oracle.residual_check(lhs, mass, rhs, xm, 2)
    assert abs(eng.residual_norm(rhs, xg, 2) - want) <= (1e-7 if cfg == "4r" else 1e-8 * want + 1e-12)
    if cfg == "5":
        # a SECOND cycle, device from the device's iterate and model from the model's: with the reference's tau = 1e-3 the iteration does not
        # contract, so this -- not a ratio of residual histories -- is the check that the device runs the same iteration at full size
        xg2 = eng.vcycle(rhs, xg)
        xm2 = M.vcycle(rhs, xm.copy())
        d2 = np.linalg.norm(xg2 - xm2) / np.linalg.norm(xm2)
        back2 = float(np.linalg.norm(lhs @ (xg2 - xm2)) / (spla.norm(lhs) * np.linalg.norm(xm2)))
        _record(config=case["name"], second_cycle_model_distance=float(d2), second_cycle_model_backward_error=back2)
        assert back2 <= 1e-12 and d2 <= 1e-6, (back2, d2)
        want2 = oracle.residual_check(lhs, mass, rhs, xm2, 2)
        assert abs(eng.residual_norm(rhs, xg2, 2) - want2) <= 1e-7 * want2 + 1e-12


def test_vcycle_is_affine(case):
    eng, rhs = case["eng"], case["rhs"]
    n = rhs.shape[0]
    rng = np.random.default_rng(7)
    b1, b2 = rhs[:, 0].copy(), rng.standard_normal(n) * np.abs(rhs[:, 0]).mean()
    x1, x2 = rng.standard_normal(n), rng.standard_normal(n)
    v = lambda b, x: eng.vcycle(b[:, None], x[:, None])[:, 0]
    y1, y2, y12, y0 = v(b1, x1), v(b2, x2), v(b1 + b2, x1 + x2), v(np.zeros(n), np.zeros(n))
    assert np.abs(y0).max() == 0.0
    # to rounding; the coarsest solve of the nearly singular Poisson operators (tau = 1e-6) amplifies rounding along the
    # constant mode by ~1/tau: the defect is a constant of relative size ~1e-9
    assert np.abs(y12 - (y1 + y2)).max() <= 1e-8 * (np.abs(y1).max() + np.abs(y2).max())


def test_hierarchy_invariants_at_full_size(case, cabi):
    """The hierarchy the constructor builds (gmg_hierarchy_build: the restatement of multigrid_solver.cpp:62-469, 975-1056) on
    the full-size workloads: what the reference's construction guarantees, checked on every level -- <= 3 parents per row,
    weights in [0, 1] summing to 1, every coarse point has children, a sample is the centre of its own cluster and (default
    barycentric weighting, not nested) clusters are what the prolongation of a sample row points at most strongly, level sizes
    fall by the FASTDISK ratio (~6 for ratio = 8) down to >= lower_bound.  (Parent-by-parent agreement with the independent
    restatement of the reference's algorithm is checked at sizes it can handle: tests/test_gpu_hierarchy.py.)"""
    import scipy.sparse as sp
    H = case["H"]
    n = case["lhs"].shape[0]
    assert len(H.U) >= (1 if case["cfg"] == "1" else 2) and len(H.samples) == len(H.nearest) == len(H.points) == len(H.U)
    for k, U in enumerate(H.U):
        nc = U.shape[1]
        assert U.shape[0] == n and 1000 <= nc < n
        assert 3.0 <= n / nc <= 12.0, (k, n, nc)
        R = sp.csr_matrix(U)
        per_row = np.diff(R.indptr)
        assert per_row.min() >= 1 and per_row.max() <= 3
        assert U.data.min() >= -1e-12 and U.data.max() <= 1 + 1e-12
        assert np.abs(np.asarray(U.sum(axis=1)).ravel() - 1.0).max() <= 1e-12
        assert np.all(np.diff(U.indptr) >= 1) and U.has_sorted_indices
        smp, near, pts = H.samples[k], H.nearest[k], H.points[k]
        assert smp.shape == (nc,) and near.shape == (n,) and pts.shape == (nc, 3)
        assert len(np.unique(smp)) == nc and smp.min() >= 0 and smp.max() < n
        assert near.min() >= 0 and near.max() < nc
        assert np.array_equal(near[smp], np.arange(nc))                      # a sample belongs to its own cluster
        assert np.all(np.bincount(near, minlength=nc) >= 1)                  # no empty cluster
        assert np.isfinite(pts).all()
        n = nc
    assert n < 8 * 1000 * 8                                                  # the coarsest level is small enough for the host LDL^T
