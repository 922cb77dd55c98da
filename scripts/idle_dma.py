"""Diagnostic: the first large DMA after a compute-only phase (python scripts/idle_dma.py a|b|c|d)."""
import sys, time
sys.path.insert(0, '.')
import bench
from gravo_mg_amd import cabi
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
v = sys.argv[1]
def timed(tag, fn):
    t = time.perf_counter(); fn(); print(f"variant {v}: {tag} {1e3 * (time.perf_counter() - t):.2f} ms")
timed("load #1", lambda: eng.load_problem(rhs, rhs))
if v == "a":
    eng.run_cycles(23, 2); timed("load #2 after 23 cycles", lambda: eng.load_problem(rhs, rhs))
elif v == "b":
    eng.run_cycles(23, 2); timed("fetch after 23 cycles", lambda: eng.fetch_solution()); timed("load #2", lambda: eng.load_problem(rhs, rhs))
elif v == "c":
    eng.run_cycles(2, 2); timed("load #2 after 2 cycles", lambda: eng.load_problem(rhs, rhs))
elif v == "d":
    time.sleep(0.03); timed("load #2 after 30 ms sleep, no cycles", lambda: eng.load_problem(rhs, rhs))
elif v == "e":
    eng.run_cycles(23, -1); timed("load #2 after 23 cycles without norm", lambda: eng.load_problem(rhs, rhs))
elif v == "f":
    eng.run_cycles(23, 2); timed("load #2 from fresh copies", lambda: eng.load_problem(rhs.copy(), rhs.copy()))
elif v == "g":
    eng.run_cycles(23, 2); timed("solve", lambda: eng.solve(rhs)); print("   solve_load", eng.timing("solve_load"), "sync", eng.timing("load_sync"))
elif v == "h":
    eng.run_cycles(23, 2); import numpy as np; big = np.empty((8_000_000,), np.float64); big[:] = 1.0
    timed("load #2 after touching a fresh 64 MB array", lambda: eng.load_problem(rhs, rhs))
elif v == "j":
    eng.run_cycles(3, 2); eng.run_cycles(20, 2); timed("solve after 3 + 20 cycles", lambda: eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)); print("   solve_load", eng.timing("solve_load"), "sync", eng.timing("load_sync"))
elif v == "k":
    eng.run_cycles(23, 2); timed("solve(tol, stop_type, max_iter) after 23 cycles", lambda: eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)); print("   solve_load", eng.timing("solve_load"), "sync", eng.timing("load_sync"))
elif v in ("m", "n"):
    import numpy as np, ctypes as C
    X = rhs.copy(order="F"); conv = np.zeros(200); it = C.c_int(); res = C.c_double()
    if v == "n":
        import gc; gc.disable()
    for rep in range(3):
        eng.run_cycles(23, 2)
        X[:] = rhs
        t = time.perf_counter()
        rc = cabi.lib().gmg_solve(eng._h, cabi._pd(rhs), cabi._pd(X), 1, 1e-4, 2, 100, C.byref(it), C.byref(res), cabi._pd(conv))
        print(f"variant {v}: direct gmg_solve with a preallocated x, rep {rep}: {1e3 * (time.perf_counter() - t):.2f} ms, load_sync {eng.timing('load_sync'):.2f}")
    for rep in range(3):
        eng.run_cycles(23, 2)
        t = time.perf_counter(); eng.solve(rhs); print(f"variant {v}: cabi solve rep {rep}: {1e3 * (time.perf_counter() - t):.2f} ms, load_sync {eng.timing('load_sync'):.2f}")
elif v == "i":
    eng.run_cycles(23, 2); c = rhs.copy()
    timed("load #2 after rhs.copy()", lambda: eng.load_problem(rhs, rhs))
timed("load #3", lambda: eng.load_problem(rhs, rhs))
