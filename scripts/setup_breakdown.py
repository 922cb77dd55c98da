import sys, time
sys.path.insert(0, '.')
import bench
from gravo_mg_amd import cabi
order = sys.argv[1] if len(sys.argv) > 1 else "natural"
H, mass, lhs, rhs = bench.build_workload(1732, 1732, order)
keys = ("reduction", "coarsest_solve", "upload", "setup_ordering", "setup_ordering_l0", "setup_ordering_l1", "setup_wait_ordering",
        "setup_device_layout", "setup_total", "setup_ordering_cached")
eng = cabi.Engine()
eng.use_hierarchy(H); eng.set_mass(mass)
for rep in range(3):          # rep 0: cold; rep 1, 2: same sparsity pattern -> cached orderings
    t = time.perf_counter(); eng.set_system(lhs); tot = time.perf_counter() - t
    print("set_system %.0f ms:" % (1e3 * tot), {k: round(eng.timing(k)) for k in keys})
    t = time.perf_counter(); x, it, res, conv = eng.solve(rhs); print("solve call %.1f ms, cycles %.1f ms, iters %d" % (1e3 * (time.perf_counter() - t), eng.timing("cycles"), it))
