"""Deterministic synthetic inputs for the V-cycle hot path (no meshes ship with the reference and there
is no network): jittered torus meshes with a cotangent Laplacian, and point clouds with a kNN graph
Laplacian (stand-in for robust_laplacian.point_cloud_laplacian, which is unavailable here).

The systems are assembled exactly the way the reference's experiment driver does
(experiments/python/comparisons.py:30-55, 75-96 and demos/smoothing.py:43-50):

    S = -cotmatrix (symmetric positive semi-definite),  M = lumped vertex areas,
    Poisson    lhs = tau*M + S,   rhs = M @ y,  y ~ N(0,1) (seed 42),  tau = 1e-6
    smoothing  lhs = M + tau*S,   rhs = M @ V,                         tau = 1e-3
    bilaplace  B = S M^-1 S  in place of S

Everything is vectorised numpy/scipy and runs at 3 M vertices in well under a minute.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def torus_mesh(n1: int, n2: int, R: float = 1.0, r: float = 0.4, jitter: float = 0.25, seed: int = 7, order: str = "natural"):
    """n1 x n2 periodic grid on a torus, two triangles per quad (valence 6 => 7 nnz per Laplacian row).

    The (u, v) parameters are jittered by `jitter` cell widths so that no two edges have identical
    length (the hierarchy's Dijkstra then has no exact ties).  Returns (V [n,3] float64, F [m,3] int32),
    area-normalised and centred like gravomg.util.normalize_area (gravomg_bindings/src/gravomg/util.py:52-55).
    order: "natural" (row-major grid order, banded matrix), "random" (worst-case permutation) or "chunks" (runs of 65 536
    consecutive vertices of the natural order, the runs themselves shuffled: local inside a run, far jumps at its borders --
    an ordering with SOME locality, below the engine's reordering trigger of mean |row - column| > n / 32)."""
    rng = np.random.default_rng(seed)
    i, j = np.meshgrid(np.arange(n1), np.arange(n2), indexing="ij")
    u = (i + jitter * (rng.random((n1, n2)) - 0.5)) * (2 * np.pi / n1)
    v = (j + jitter * (rng.random((n1, n2)) - 0.5)) * (2 * np.pi / n2)
    x = (R + r * np.cos(v)) * np.cos(u)
    y = (R + r * np.cos(v)) * np.sin(u)
    z = r * np.sin(v)
    V = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    idx = (i * n2 + j)
    ip = ((i + 1) % n1) * n2 + j
    jp = i * n2 + (j + 1) % n2
    ipjp = ((i + 1) % n1) * n2 + (j + 1) % n2
    F = np.concatenate([
        np.stack([idx.ravel(), ip.ravel(), ipjp.ravel()], axis=1),
        np.stack([idx.ravel(), ipjp.ravel(), jp.ravel()], axis=1),
    ]).astype(np.int32)
    if order in ("random", "chunks"):
        if order == "random":
            perm = rng.permutation(V.shape[0])            # new -> old
        else:
            run = 65536
            starts = np.arange(0, V.shape[0], run)
            perm = np.concatenate([np.arange(s0, min(s0 + run, V.shape[0])) for s0 in starts[rng.permutation(len(starts))]])
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.shape[0])
        V = V[perm]
        F = inv[F].astype(np.int32)
    elif order != "natural":
        raise ValueError(order)
    V = normalize_area(V, F)
    return V, F


def face_area(V, F):
    e1 = V[F[:, 1]] - V[F[:, 0]]
    e2 = V[F[:, 2]] - V[F[:, 0]]
    return 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)


def normalize_area(V, F):
    V = V / np.sqrt(face_area(V, F).sum())
    return V - V.mean(axis=0, keepdims=True)


def cotan_laplacian(V, F):
    """S = -cotmatrix(V, F) (positive semi-definite stiffness, CSC) and lumped mass diagonal
    (one third of the incident triangle areas).  The reference uses igl.cotmatrix and igl.massmatrix
    (VORONOI) -- experiments/python/comparisons.py:41-42; both are lumped diagonal masses."""
    n = V.shape[0]
    rows, cols, vals = [], [], []
    for a in range(3):
        i0, i1, i2 = F[:, a], F[:, (a + 1) % 3], F[:, (a + 2) % 3]
        # angle at i0 is opposite to edge (i1, i2)
        e1 = V[i1] - V[i0]
        e2 = V[i2] - V[i0]
        cot = np.einsum("ij,ij->i", e1, e2) / np.linalg.norm(np.cross(e1, e2), axis=1)
        w = 0.5 * cot
        rows += [i1, i2, i1, i2]
        cols += [i2, i1, i1, i2]
        vals += [-w, -w, w, w]
    S = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
    S.sum_duplicates()
    S.sort_indices()
    area = face_area(V, F)
    mass = np.zeros(n)
    for a in range(3):
        mass += np.bincount(F[:, a], weights=area / 3.0, minlength=n)
    return S, mass


def neighbors_from_stiffness(S) -> np.ndarray:
    """(n, kmax) int32 table: row i = ascending column ids of row i of S (the vertex itself included,
    from the diagonal), padded with -1.  Same output as gravomg.util.neighbors_from_stiffness
    (gravomg_bindings/src/gravomg/util.py:4-8,36-44) for its CSC input, but also correct for any other
    scipy format (the upstream helper silently mis-sorts CSR input)."""
    S = sp.csr_matrix(S)
    S.sort_indices()
    n = S.shape[0]
    deg = np.diff(S.indptr)
    kmax = int(deg.max())
    neigh = -np.ones((n, kmax), dtype=np.int32)
    row = np.repeat(np.arange(n), deg)
    pos = np.arange(S.nnz) - np.repeat(S.indptr[:-1], deg)
    neigh[row, pos] = S.indices
    return neigh


def knn_graph_laplacian(P, k: int = 8):
    """Symmetric kNN graph Laplacian with positive Gaussian-ish weights and a lumped 'mass' from the
    local sampling density; stand-in for robust_laplacian.point_cloud_laplacian (comparisons.py:43-44)."""
    from scipy.spatial import cKDTree

    n = P.shape[0]
    tree = cKDTree(P)
    dist, nb = tree.query(P, k + 1)
    dist, nb = dist[:, 1:], nb[:, 1:]
    h2 = np.mean(dist[:, -1] ** 2)
    w = np.exp(-(dist ** 2) / h2) + 1e-3
    rows = np.repeat(np.arange(n), k)
    W = sp.coo_matrix((w.ravel(), (rows, nb.ravel())), shape=(n, n)).tocsr()
    W = W.maximum(W.T)                      # undirected
    W.setdiag(0)
    W.eliminate_zeros()
    deg = np.asarray(W.sum(axis=1)).ravel()
    S = (sp.diags(deg) - W).tocsc()
    S.sort_indices()
    mass = np.pi * dist[:, -1] ** 2 / k     # area per sample ~ disc to the k-th neighbour / k
    mass = mass / mass.sum()
    return S, mass


def torus_points(n: int, R: float = 1.0, r: float = 0.4, noise: float = 0.0, seed: int = 11):
    rng = np.random.default_rng(seed)
    u = rng.random(n) * 2 * np.pi
    v = rng.random(n) * 2 * np.pi
    P = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], axis=1)
    if noise > 0:
        P = P + noise * rng.standard_normal(P.shape)
    P = P - P.mean(axis=0, keepdims=True)
    return P * (0.5 / np.abs(P).max())      # gravomg.util.normalize_bounding_box (util.py:57-60)


def poisson_system(S, mass, tau: float = 1e-6, seed: int = 42, d: int = 1):
    """lhs = tau*M + S, rhs = M @ y, y ~ N(0,1)   (comparisons.py:75-76, 85-96; comparison_poisson.sh:2-4)."""
    n = S.shape[0]
    lhs = (sp.diags(mass) * tau + S).tocsc()
    lhs.sort_indices()
    y = np.random.default_rng(seed).standard_normal((n, d))
    rhs = mass[:, None] * y
    return lhs, np.asfortranarray(rhs)


def smoothing_system(S, mass, V, tau: float = 1e-3):
    """lhs = M + tau*S, rhs = M @ V (n x 3)   (demos/smoothing.py:43-50; comparisons.py:78)."""
    lhs = (sp.diags(mass) + tau * S).tocsc()
    lhs.sort_indices()
    rhs = mass[:, None] * V
    return lhs, np.asfortranarray(rhs)


def bilaplacian(S, mass):
    """B = S M^-1 S (comparisons.py:54)."""
    B = (S @ sp.diags(1.0 / mass) @ S).tocsc()
    B.sort_indices()
    return B


def open_cylinder_mesh(n1: int, n2: int, radius: float = 1.0, height: float = 2.0, jitter: float = 0.25, seed: int = 5):
    """A surface WITH BOUNDARY: n1 (around, periodic) x n2 (along the axis, open at both ends) grid on a cylinder, two triangles
    per quad.  The two rims are boundary loops (valence-4 rows with no Dirichlet condition: the cotangent Laplacian's natural
    boundary), everything else is valence 6.  Area-normalised and centred like the torus meshes."""
    rng = np.random.default_rng(seed)
    i, j = np.meshgrid(np.arange(n1), np.arange(n2), indexing="ij")
    u = (i + jitter * (rng.random((n1, n2)) - 0.5)) * (2 * np.pi / n1)
    t = (j + jitter * (rng.random((n1, n2)) - 0.5) * ((j > 0) & (j < n2 - 1))) * (height / (n2 - 1))
    V = np.stack([(radius * np.cos(u)).ravel(), (radius * np.sin(u)).ravel(), t.ravel()], axis=1)
    ii, jj = np.meshgrid(np.arange(n1), np.arange(n2 - 1), indexing="ij")
    a = (ii * n2 + jj).ravel(); b = (((ii + 1) % n1) * n2 + jj).ravel(); c = (((ii + 1) % n1) * n2 + jj + 1).ravel(); d = (ii * n2 + jj + 1).ravel()
    F = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)]).astype(np.int32)
    return normalize_area(V, F), F


def sheared_torus_mesh(n1: int, n2: int, shear: float = 2.5, **kw):
    """A torus mesh with BADLY SHAPED triangles: the grid of torus_mesh with its quads split along the LONG diagonal of a sheared
    parameter lattice, which makes every triangle obtuse (largest angle ~ atan-dependent on `shear`; 2.5 gives ~130 degrees).
    Obtuse angles have negative cotangents, so the stiffness matrix S = -cotmatrix gets POSITIVE off-diagonal entries: it stays
    symmetric positive semi-definite but is no longer an M-matrix -- the class for which over-relaxed and block-hybrid sweeps
    have no textbook guarantee."""
    V, F = torus_mesh(n1, n2, **kw)
    # re-triangulate: vertex (i, j) connects to (i+1, j + s) instead of (i+1, j): a lattice sheared by s cells
    s_cells = int(round(shear))
    i, j = np.meshgrid(np.arange(n1), np.arange(n2), indexing="ij")
    idx = (i * n2 + j).ravel()
    right = (((i + 1) % n1) * n2 + (j + s_cells) % n2).ravel()
    up = (i * n2 + (j + 1) % n2).ravel()
    right_up = (((i + 1) % n1) * n2 + (j + s_cells + 1) % n2).ravel()
    F2 = np.concatenate([np.stack([idx, right, right_up], axis=1), np.stack([idx, right_up, up], axis=1)]).astype(np.int32)
    return normalize_area(V, F2), F2


def sphere_mesh(n: int, seed: int = 3, order: str = "spatial"):
    """Irregular triangle mesh: n random points on the unit sphere, triangulated by their convex hull (valence 3..12,
    mean 6).  order "spatial" sorts the vertices along a Morton curve of their coordinates (scanner-like locality);
    "random" keeps the random order.  Area-normalised and centred like the torus meshes."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((n, 3))
    P /= np.linalg.norm(P, axis=1, keepdims=True)
    if order == "spatial":
        q = np.clip(((P + 1.0) * 0.5 * 1023).astype(np.int64), 0, 1023)
        key = np.zeros(n, np.int64)
        for b in range(10):
            for a in range(3):
                key |= ((q[:, a] >> b) & 1) << (3 * b + a)
        P = P[np.argsort(key, kind="stable")]
    elif order != "random":
        raise ValueError(order)
    hull = ConvexHull(P)
    F = hull.simplices.astype(np.int32)
    # consistent outward orientation (hull simplices are unordered)
    c = P[F].mean(axis=1)
    nrm = np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]])
    flip = np.einsum("ij,ij->i", nrm, c) < 0
    F[flip] = F[flip][:, [0, 2, 1]]
    return normalize_area(P, F), F


def baseline_config(cfg: str):
    """BASELINE.json's configs as synthetic stand-ins at FULL size (no mesh ships with the reference and there
    is no network).  Returns (name, positions, S, mass, lhs, rhs).  "4r": config 4 in random vertex order; "5": the
    reference's smoothing parameter tau = 1e-3 (comparison_smoothing.sh:2-3), "5b": tau = 1e-9, for which the reference
    iteration contracts at this size; "6": an irregular-valence mesh (not in BASELINE.json)."""
    if cfg == "1":      # demos/smoothing.py call pattern: M + 1e-3 S, rhs = M V (n x 3), ~36 k vertices
        V, F = torus_mesh(190, 190)
        S, mass = cotan_laplacian(V, F)
        lhs, rhs = smoothing_system(S, mass, V)
        return "cfg1 torus 190x190 smoothing d=3", V, S, mass, lhs, rhs
    if cfg == "2":      # ~720 k cotan Poisson
        V, F = torus_mesh(850, 850)
        S, mass = cotan_laplacian(V, F)
        lhs, rhs = poisson_system(S, mass)
        return "cfg2 torus 850x850 Poisson d=1", V, S, mass, lhs, rhs
    if cfg == "3":      # ~2 M point cloud, kNN graph Laplacian (stand-in for robust_laplacian)
        P = torus_points(2_000_000, noise=0.0005)
        S, mass = knn_graph_laplacian(P, 8)
        lhs, rhs = poisson_system(S, mass)
        return "cfg3 point cloud 2M kNN(8) Poisson d=1", P, S, mass, lhs, rhs
    if cfg in ("4", "4r"):   # ~3 M mesh Poisson (the bench workload), natural / random vertex order
        V, F = torus_mesh(1732, 1732, order="random" if cfg == "4r" else "natural")
        S, mass = cotan_laplacian(V, F)
        lhs, rhs = poisson_system(S, mass)
        return f"cfg4 torus 1732x1732 Poisson d=1 ({'random' if cfg == '4r' else 'natural'} vertex order)", V, S, mass, lhs, rhs
    if cfg == "4s":     # the reference's real call pattern at the headline size: smoothing M + 1e-3 S, rhs = M V (n x 3), on the 3 M mesh
        V, F = torus_mesh(1732, 1732)
        S, mass = cotan_laplacian(V, F)
        lhs, rhs = smoothing_system(S, mass, V)
        return "cfg4s torus 1732x1732 smoothing d=3", V, S, mass, lhs, rhs
    if cfg in ("5", "5b"):   # Bilaplacian data smoothing M + tau S M^-1 S on the ~3 M mesh (mixed precision: fp32 inner V-cycle)
        tau = 1e-3 if cfg == "5" else 1e-9
        V, F = torus_mesh(1732, 1732)
        S, mass = cotan_laplacian(V, F)
        lhs, rhs = smoothing_system(bilaplacian(S, mass), mass, V[:, :1], tau=tau)
        return f"cfg5 torus 1732x1732 Bilaplacian smoothing tau={tau:g} d=1", V, S, mass, lhs, rhs
    if cfg == "6":      # irregular-valence mesh (random points on a sphere, hull triangulation): 7 colours, ragged rows
        V, F = sphere_mesh(1_000_000)
        S, mass = cotan_laplacian(V, F)
        lhs, rhs = poisson_system(S, mass)
        return "cfg6 irregular sphere 1M Poisson d=1 (valence 3..13)", V, S, mass, lhs, rhs
    raise ValueError(cfg)
