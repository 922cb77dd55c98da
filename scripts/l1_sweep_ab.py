"""Level-1 block sweep on the 3 M-vertex bench workload: time per sweep for engine variants / debug modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
V, F = meshgen.torus_mesh(1732, 1732)
S, mass = meshgen.cotan_laplacian(V, F)
lhs, rhs = meshgen.poisson_system(S, mass)
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
variants = [a for a in sys.argv[1:]] or ["", "block_ep=0"]
for v in variants:
    kw = dict(a.split('=') for a in v.split(',') if a and not a.startswith("DBG"))
    kw = {k: int(x) for k, x in kw.items()}
    eng = cabi.Engine(**kw)
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    for dbg in ["0"]:
        line = f"{v or 'default':30s} dbg={dbg:3s}"
        for d in (1, 3):
            t_ms, _ = eng.bench_kernel(0, 1, d, 50)
            line += f" | d={d} L1 sweep {1e3 * t_ms:.1f} us"
        if dbg == "0":
            eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
            t = time.perf_counter(); eng.run_cycles(20, 2); line += f" | cycle {50 * (time.perf_counter() - t):.3f} ms"
        print(line, flush=True)
    eng.close()
