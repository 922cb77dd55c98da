"""gmg_fetch_solution / gmg_load_problem at the bench workload with pre-touched host arrays: what the staged transfers cost.
  python scripts/fetch_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gravo_mg_amd import cabi
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
eng = cabi.Engine()
eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
B = np.asfortranarray(rhs.reshape(len(rhs), -1))
eng.load_problem(B, B); eng.run_cycles(4, 2)
X = np.zeros_like(B, order="F")
l = cabi.lib()
for name, fn in (("fetch_solution", lambda: l.gmg_fetch_solution(eng._h, cabi._pd(X))),
                 ("load_problem(b, x0 = b)", lambda: l.gmg_load_problem(eng._h, cabi._pd(B), cabi._pd(B), B.shape[1]))):
    ms = []
    for _ in range(12):
        t = time.perf_counter(); rc = fn(); ms.append(1e3 * (time.perf_counter() - t)); assert rc == 0
    print(f"{name}: {[round(v, 3) for v in ms]} ms for {B.nbytes / 1e6:.1f} MB", flush=True)
for threads in (1, 4, 16):
    t = time.perf_counter()
    for _ in range(10): np.copyto(X, B)
    print("numpy copy of the same array:", round(1e2 * (time.perf_counter() - t), 3), "ms")
    break
