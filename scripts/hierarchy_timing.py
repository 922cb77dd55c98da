"""Phases of the host hierarchy builder on the bench mesh (the reference's hierarchyTiming keys)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi, meshgen
order = sys.argv[1] if len(sys.argv) > 1 else "natural"
V, F = meshgen.torus_mesh(1732, 1732, order=order)
S, mass = meshgen.cotan_laplacian(V, F)
nb = meshgen.neighbors_from_stiffness(S)
for rep in range(2):
    t = time.perf_counter(); H = cabi.Hierarchy(V, nb); dt = time.perf_counter() - t
    keys = ["hierarchy", "sampling", "cluster", "next_neighborhood", "next_positions", "triangle_finding", "triangle_selection", "prepare", "edge_length", "assemble", "levels"]
    print(order, round(dt, 3), {k: round(H.timing(k), 1) for k in keys})
