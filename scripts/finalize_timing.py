import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(1732, 1732)
S, mass = meshgen.cotan_laplacian(V, F)
nb = meshgen.neighbors_from_stiffness(S)
t = time.perf_counter(); H = cabi.Hierarchy(V, nb); print("Hierarchy", time.perf_counter() - t)
t = time.perf_counter(); eng = cabi.Engine(); print("Engine()", time.perf_counter() - t)
t = time.perf_counter(); eng.use_hierarchy(H); print("use_hierarchy (set U + finalize)", time.perf_counter() - t)
t = time.perf_counter(); eng.set_mass(mass); print("set_mass", time.perf_counter() - t)
for k in ("finalize_ms", "patches_ms", "transfers_ms"):
    try: print(k, eng.timing(k))
    except Exception as e: pass
