"""GPU-suite copy of the hierarchy parity check: the product's hierarchy builder (gmg_hierarchy_build, host_hierarchy.hpp)
against the independent pure-Python restatement of the reference's constructProlongation (oracle/hierarchy_restatement.py) --
same levels, same parents, same weights -- on the five BASELINE workload TYPES at sizes the Python version handles, and the
engine consuming that hierarchy: same Galerkin operators as the oracle given the restatement's U, same solve."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _workload(kind):
    from gravo_mg_amd import meshgen
    if kind == "pointcloud":
        V = meshgen.torus_points(2500, noise=0.003)
        S, mass = meshgen.knn_graph_laplacian(V, 8)
        lhs, rhs = meshgen.poisson_system(S, mass)
    else:
        V, F = meshgen.torus_mesh(48, 40, order="random" if kind == "poisson-random" else "natural")
        S, mass = meshgen.cotan_laplacian(V, F)
        if kind == "smoothing-d3":
            lhs, rhs = meshgen.smoothing_system(S, mass, V)
        elif kind == "bilaplacian":
            lhs, rhs = meshgen.smoothing_system(meshgen.bilaplacian(S, mass), mass, V[:, :1], tau=1e-9)
        else:
            lhs, rhs = meshgen.poisson_system(S, mass)
    return V, meshgen.neighbors_from_stiffness(S), mass, lhs, rhs


@pytest.mark.parametrize("kind", ["smoothing-d3", "poisson", "pointcloud", "poisson-random", "bilaplacian"])
def test_builder_matches_restatement_and_engine_consumes_it(cabi, oracle, kind):
    from oracle import hierarchy_restatement as ref
    V, neigh, mass, lhs, rhs = _workload(kind)
    H = cabi.Hierarchy(V, neigh, lower_bound=12)
    want = ref.build(V, neigh, ratio=8.0, lower_bound=12)
    assert len(H.U) == len(want) >= 2
    for k, (a, b) in enumerate(zip(H.U, want)):
        a, b = sp.csc_matrix(a), sp.csc_matrix(b)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), f"level {k}: different parents"
        np.testing.assert_allclose(a.data, b.data, rtol=1e-9, atol=1e-12)
    # the engine on the product's hierarchy vs the oracle on the restatement's: same Galerkin operators, same solve
    eng = cabi.Engine()
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    O = oracle.Hierarchy(want, mass)
    O.set_system(lhs)
    for k in range(1, len(want) + 1):
        A, Ao = eng.level_operator(k), O.level_operator(k)
        assert abs(A - Ao).max() <= 1e-9 * abs(Ao).max()
    tol = 1e-2 if kind == "bilaplacian" else 1e-4
    x, it, res, _ = eng.solve(rhs, tol=tol, max_iter=100)
    xo, ito, reso, _ = O.solve(rhs, tol=tol, max_iter=100)
    assert res <= tol and reso <= tol and it <= ito + 2
    m = mass[:, None]
    assert np.sqrt((m * (x - xo) ** 2).sum() / (m * xo ** 2).sum()) <= 20 * tol


@pytest.mark.parametrize("kind,weighting,nested", [("torus", 0, False), ("torus-random", 0, False), ("sphere", 0, False), ("pointcloud", 0, False),
                                                    ("torus", 2, False), ("torus", 1, True)])
def test_device_selection_stage_gives_the_host_bits(cabi, kind, weighting, nested):
    """The per-point parent selection (multigrid_solver.cpp:291-452) runs on the GPU for levels of >= 200 k points
    (csrc/hierarchy_kernels.hip.hpp); U must be bit-identical to the host loop's (gmg_hierarchy_options::use_device = 0), whatever the mesh."""
    import os
    from gravo_mg_amd import meshgen
    if kind.startswith("torus"):
        V, F = meshgen.torus_mesh(520, 500, order="random" if kind.endswith("random") else "natural")
        S, _ = meshgen.cotan_laplacian(V, F)
        neigh = meshgen.neighbors_from_stiffness(S)
    elif kind == "sphere":
        V, F = meshgen.sphere_mesh(230_000)
        S, _ = meshgen.cotan_laplacian(V, F)
        neigh = meshgen.neighbors_from_stiffness(S)
    else:
        V = meshgen.torus_points(240_000, noise=0.002)
        S, _ = meshgen.knn_graph_laplacian(V, 8)
        neigh = meshgen.neighbors_from_stiffness(S)
    Hh = cabi.Hierarchy(V, neigh, weighting=weighting, nested=nested, use_device=False)
    Hd = cabi.Hierarchy(V, neigh, weighting=weighting, nested=nested)
    assert Hh.timing("selection_on_device") == 0.0 and Hd.timing("selection_on_device") >= 1.0
    assert len(Hh.U) == len(Hd.U) >= 2
    for a, b in zip(Hh.U, Hd.U):
        assert a.shape == b.shape
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
        assert np.array_equal(a.data, b.data)            # bitwise
