"""The product's C++ hierarchy builder (gmg_hierarchy_build, host_hierarchy.hpp) against an independent pure-Python
restatement of the reference's constructProlongation (oracle/hierarchy_restatement.py): same levels, same sparsity
pattern, same weights.  CPU only, small inputs (the Python version is slow by design)."""
import numpy as np
import pytest
import scipy.sparse as sp

from gravo_mg_amd import meshgen


def _inputs(kind):
    if kind == "torus":
        V, F = meshgen.torus_mesh(30, 24)
        S, _ = meshgen.cotan_laplacian(V, F)
    elif kind == "torus-random":
        V, F = meshgen.torus_mesh(26, 22, order="random")
        S, _ = meshgen.cotan_laplacian(V, F)
    else:
        V = meshgen.torus_points(600, noise=0.003)
        S, _ = meshgen.knn_graph_laplacian(V, 7)
    return V, meshgen.neighbors_from_stiffness(S)


@pytest.mark.parametrize("kind,lower_bound,ratio", [("torus", 15, 8.0), ("torus-random", 8, 8.0), ("pointcloud", 20, 8.0), ("torus", 10, 20.0)])
def test_cpp_builder_matches_python_restatement(cabi, kind, lower_bound, ratio):
    from oracle import hierarchy_restatement as ref
    V, neigh = _inputs(kind)
    got = cabi.Hierarchy(V, neigh, lower_bound=lower_bound, ratio=ratio).U
    want = ref.build(V, neigh, ratio=ratio, lower_bound=lower_bound)
    assert len(got) == len(want) >= 1
    kinds = {1: 0, 2: 0, 3: 0}
    for k, (a, b) in enumerate(zip(got, want)):
        a, b = sp.csc_matrix(a), sp.csc_matrix(b)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), f"level {k}: different parents"
        np.testing.assert_allclose(a.data, b.data, rtol=1e-9, atol=1e-12)
        for cnt in np.diff(sp.csr_matrix(a).indptr):
            kinds[int(cnt)] += 1
    assert kinds[3] > 0 and kinds[2] > 0          # triangle rows and edge/two-parent rows both occur


def test_fine_order_is_the_breadth_first_order_of_the_point_graph(cabi):
    """gmg_hierarchy_build's by-product for inputs without locality (gmg_hierarchy_get_fine_order): the order in which a sequential
    breadth-first search over `neigh` from point 0 dequeues the points -- checked against a plain Python BFS; a locally numbered
    input gets none."""
    V, F = meshgen.torus_mesh(340, 340, order="random")
    S, _ = meshgen.cotan_laplacian(V, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(V, neigh, lower_bound=4000)
    fo = H.fine_order
    n = V.shape[0]
    assert fo is not None and fo.shape == (n,)
    seen = np.zeros(n, bool); want = []
    for s0 in range(n):
        if seen[s0]:
            continue
        seen[s0] = True; want.append(s0); h = len(want) - 1
        while h < len(want):
            for w in neigh[want[h]]:
                if w < 0:
                    break
                if not seen[w]:
                    seen[w] = True; want.append(int(w))
            h += 1
    assert np.array_equal(fo, np.array(want, dtype=np.int32))
    V2, F2 = meshgen.torus_mesh(300, 300)
    S2, _ = meshgen.cotan_laplacian(V2, F2)
    assert cabi.Hierarchy(V2, meshgen.neighbors_from_stiffness(S2), lower_bound=4000).fine_order is None


@pytest.mark.parametrize("kind,ratio", [("torus", 8.0), ("torus-random", 8.0), ("sphere", 8.0), ("pointcloud", 8.0), ("open-cylinder", 8.0),
                                        ("sphere", 1.3), ("pointcloud", 1.3), ("torus", 27.0)])
def test_clustering_sweep_after_the_disk_sampler_only_resets_the_samples(cabi, kind, ratio):
    """multigrid_solver.cpp:1015-1056 after :975-1013: the multi-source Dijkstra starts from distances the sampler seeded with
    exactly the candidates the sweep would offer, so it cannot lower anything (argument in host_hierarchy.hpp::voronoi_dijkstra) and
    the builder skips the heap.  gmg_hierarchy_options::full_clustering runs the sweep as the reference writes it: same owners for every
    point of every level, same hierarchy bit for bit -- on regular, randomly numbered, irregular, open and point-cloud inputs, with
    small and large sampling radii."""
    import os
    if kind.startswith("torus"):
        V, F = meshgen.torus_mesh(150, 140, order="random" if kind.endswith("random") else "natural")
    elif kind == "sphere":
        V, F = meshgen.sphere_mesh(20_000)
    elif kind == "open-cylinder":
        V, F = meshgen.open_cylinder_mesh(150, 130)
    if kind == "pointcloud":
        V = meshgen.torus_points(20_000, noise=0.002)
        S, _ = meshgen.knn_graph_laplacian(V, 8)
    else:
        S, _ = meshgen.cotan_laplacian(V, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    Hf = cabi.Hierarchy(V, neigh, ratio=ratio, lower_bound=50, full_clustering=True)
    Hs = cabi.Hierarchy(V, neigh, ratio=ratio, lower_bound=50)
    assert len(Hf.U) == len(Hs.U) >= 2
    for a, b in zip(Hf.nearest, Hs.nearest):
        assert np.array_equal(a, b)
    for a, b in zip(Hf.U, Hs.U):
        assert a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)


def test_a_neighbour_behind_a_gap_in_a_table_row_makes_the_full_sweep_run(cabi):
    """The sampler stops at a row's first -1, the Dijkstra sweep skips over it: with such a row the shortcut's argument does not hold,
    and the builder must fall back to the sweep (same result as forcing it)."""
    import os
    V, neigh = _inputs("torus")
    neigh = np.ascontiguousarray(np.concatenate([neigh[:, :2], -np.ones((len(neigh), 1), neigh.dtype), neigh[:, 2:]], axis=1))
    Hf = cabi.Hierarchy(V, neigh, lower_bound=15, full_clustering=True)
    Hs = cabi.Hierarchy(V, neigh, lower_bound=15)
    assert len(Hf.U) == len(Hs.U) >= 1
    for a, b in zip(Hf.nearest, Hs.nearest):
        assert np.array_equal(a, b)
    # (with the gap the sweep does real work: points reached only through the hidden neighbours change owner)
    V0, n0 = _inputs("torus")
    assert not np.array_equal(cabi.Hierarchy(V0, n0, lower_bound=15).nearest[0], Hs.nearest[0])
