"""Per-cycle parity of the DEFAULT engine at mid size (330 x 330 = 108 900 vertices, three transfer levels): consecutive
V-cycles against the model assembled from the oracle's operators with the device's orderings (tests/vcycle_model.py) --
level-0 multicolour SOR, block-hybrid sweeps on the Galerkin levels, host LDL^T.  Same iteration, so the iterates agree to
rounding cycle by cycle, whether or not the iteration contracts: this is the tight check behind BASELINE config 5
(Bilaplacian, tau = 1e-3: the reference iteration does not contract from x0 = rhs at the sizes tested).

Tolerances: relative 2-norm of the iterate difference after each cycle (both sides restart from the model's iterate, so
rounding does not accumulate): <= 1e-10 for the smoothing system, <= 1e-9 for the Bilaplacian ones (fourth-order operator,
condition number ~ n^2: measured 1.2e-10); the Poisson systems
(tau*M + S, tau = 1e-6, ||x||/||b|| ~ 1e8) get the backward-error form ||A dx|| <= 1e-12 ||A|| ||x|| and 1e-6 forward.
The fp32 inner cycle (mixed precision) is held to 2e-5 of the fp64 model."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


def _problem(cabi, kind):
    from gravo_mg_amd import meshgen
    V, F = meshgen.torus_mesh(330, 330)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=400)
    if kind == "poisson":
        lhs, rhs = meshgen.poisson_system(S, mass)
    elif kind == "smoothing-d3":
        lhs, rhs = meshgen.smoothing_system(S, mass, V)
    else:
        lhs, rhs = meshgen.smoothing_system(meshgen.bilaplacian(S, mass), mass, V[:, :1], tau=float(kind.split(":")[1]))
    return H, mass, lhs, rhs


@pytest.mark.parametrize("lanes", [0, 1], ids=["default", "big-level-kernels"])
@pytest.mark.parametrize("kind", ["poisson", "smoothing-d3", "bilaplacian:1e-3", "bilaplacian:1e-9"])
def test_default_engine_matches_model_cycle_by_cycle(cabi, oracle, kind, lanes):
    """lanes = 1 forces the layout and sweep kernel of levels with >= 262 144 rows (one lane per row, entry-parallel block
    sweep) onto this problem's 18 k-row level 1, so the kernels of the 3 M-vertex configuration get the same check."""
    from tests.vcycle_model import VcycleModel
    H, mass, lhs, rhs = _problem(cabi, kind)
    eng = cabi.Engine(block_lanes=lanes)
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    assert eng.num_levels >= 2 and eng.level_blocks(1) is not None and eng.level_blocks(0) is None      # the default layout
    M = VcycleModel(eng, H.U, mass, lhs, oracle, eng.gs_omega)
    nA = spla.norm(lhs)
    x = rhs.copy()
    residues = []
    for cyc in range(3):
        xg = eng.vcycle(rhs, x)
        xm = M.vcycle(rhs, x)
        d = np.linalg.norm(xg - xm) / np.linalg.norm(xm)
        if kind == "poisson":
            assert np.linalg.norm(lhs @ (xg - xm)) <= 1e-12 * nA * np.linalg.norm(xm), (cyc, d)
            assert d <= 1e-6, (cyc, d)
        else:
            assert d <= (1e-9 if kind.startswith("bilaplacian") else 1e-10), (cyc, d)
        # the residue the device reports for its iterate is the oracle's residualCheck of it
        residues.append(oracle.residual_check(lhs, mass, rhs, xm, 2))
        # (the Poisson iterates are ~1e8 x the right-hand side, so A x - b cancels eight digits on either side: 1e-7 there)
        assert abs(eng.residual_norm(rhs, xg, 2) - oracle.residual_check(lhs, mass, rhs, xg, 2)) <= (1e-7 if kind == "poisson" else 1e-9 * residues[-1] + 1e-12)
        x = xm
    if kind == "bilaplacian:1e-3":
        assert residues[0] > 1.0            # the regime the test is for: far from converged, not contracting to 1e-4


@pytest.mark.parametrize("kind", ["bilaplacian:1e-3", "smoothing-d3"])
def test_mixed_precision_cycle_matches_model(cabi, oracle, kind):
    """BASELINE config 5: fp32 inner V-cycle, fp64 defect and correction.  One cycle == the fp64 model to fp32 rounding."""
    from tests.vcycle_model import VcycleModel
    H, mass, lhs, rhs = _problem(cabi, kind)
    ref = cabi.Engine()
    ref.use_hierarchy(H); ref.set_mass(mass); ref.set_system(lhs)
    mix = cabi.Engine(inner_precision=1)
    mix.use_hierarchy(H); mix.set_mass(mass); mix.set_system(lhs)
    M = VcycleModel(ref, H.U, mass, lhs, oracle, ref.gs_omega)
    x = rhs.copy()
    for cyc in range(2):
        xg = mix.vcycle(rhs, x)
        xm = M.vcycle(rhs, x)
        assert np.linalg.norm(xg - xm) <= 2e-5 * np.linalg.norm(xm), cyc
        x = xm
