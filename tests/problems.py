"""Seeded synthetic problems shared by the tests (built with the product's host-side generator and
hierarchy builder; the oracle consumes the same U_k so both sides see identical inputs)."""
from __future__ import annotations

import functools

import numpy as np

from gravo_mg_amd import cabi, meshgen


class Problem:
    def __init__(self, V, S, mass, U, lhs, rhs, name):
        self.V, self.S, self.mass, self.U, self.lhs, self.rhs, self.name = V, S, mass, U, lhs, rhs, name

    @property
    def n(self):
        return self.lhs.shape[0]


@functools.lru_cache(maxsize=None)
def torus_problem(n1=48, n2=40, kind="poisson", lower_bound=60, order="natural", d=1, seed=42):
    V, F = meshgen.torus_mesh(n1, n2, order=order)
    S, mass = meshgen.cotan_laplacian(V, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(V, neigh, lower_bound=lower_bound)
    if kind == "poisson":
        lhs, rhs = meshgen.poisson_system(S, mass, seed=seed, d=d)
    elif kind == "smoothing":
        lhs, rhs = meshgen.smoothing_system(S, mass, V)
    elif kind == "bilaplacian":
        lhs, rhs = meshgen.poisson_system(meshgen.bilaplacian(S, mass), mass, tau=1.0, seed=seed, d=d)
    else:
        raise ValueError(kind)
    return Problem(V, S, mass, H.U, lhs, rhs, f"torus{n1}x{n2}-{kind}-{order}")


@functools.lru_cache(maxsize=None)
def pointcloud_problem(n=3000, k=8, lower_bound=80):
    P = meshgen.torus_points(n, noise=0.002)
    S, mass = meshgen.knn_graph_laplacian(P, k)
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(P, neigh, lower_bound=lower_bound)
    lhs, rhs = meshgen.poisson_system(S, mass)
    return Problem(P, S, mass, H.U, lhs, rhs, f"pointcloud{n}")


@functools.lru_cache(maxsize=None)
def sphere_problem(n=6000, lower_bound=80, order="spatial"):
    """Irregular-valence mesh (random points on a sphere, convex-hull triangulation): 6-8 colours, ragged rows."""
    V, F = meshgen.sphere_mesh(n, order=order)
    S, mass = meshgen.cotan_laplacian(V, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    H = cabi.Hierarchy(V, neigh, lower_bound=lower_bound)
    lhs, rhs = meshgen.poisson_system(S, mass)
    return Problem(V, S, mass, H.U, lhs, rhs, f"sphere{n}-{order}")


def permuted_system(A, new2old):
    """P A P^T restricted to the real rows of a device ordering (padding rows dropped)."""
    order = new2old[new2old >= 0]
    return A.tocsr()[order][:, order].tocsc(), order
