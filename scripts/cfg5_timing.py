"""BASELINE config 5 (3 M Bilaplacian, tau = 1e-3) and the d = 3 smoothing system at 3 M vertices: ms per cycle for fp64 / mixed precision,
with the unpadded block sweep (default) and the SELL / block-CSR sweeps it replaced (block_ep=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi, meshgen
name, V, S, mass, lhs, rhs = meshgen.baseline_config("5")
H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S))
lhs3, rhs3 = meshgen.smoothing_system(S, mass, V)
for label, A, b in ((name, lhs, rhs), ("smoothing d=3 at 3 M vertices", lhs3, rhs3)):
    for kw in (dict(), dict(block_ep=0), dict(inner_precision=1), dict(inner_precision=1, block_ep=0)):
        eng = cabi.Engine(**kw)
        eng.use_hierarchy(H); eng.set_mass(mass)
        t = time.perf_counter(); eng.set_system(A); ts = 1e3 * (time.perf_counter() - t)
        eng.load_problem(b, b); eng.run_cycles(3, 2)
        t = time.perf_counter(); r = eng.run_cycles(20, 2); ms = 50 * (time.perf_counter() - t)
        t1, _ = eng.bench_kernel(0, 1, b.shape[1], 30)
        print(f"{label:60s} {str(kw):45s} {ms:.3f} ms/cycle  set_system {ts:.0f} ms  L1 sweep {1e3 * t1:.1f} us  res {r[-1]:.4e}", flush=True)
        eng.close()
