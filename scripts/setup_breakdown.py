import sys, time
sys.path.insert(0, '.')
import bench
from gravo_mg_amd import cabi
order = sys.argv[1] if len(sys.argv) > 1 else "natural"
N1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1732
H, mass, lhs, rhs = bench.build_workload(N1, N1, order)
keys = ("reduction", "coarsest_solve", "upload", "setup_ordering", "setup_ordering_l0", "setup_ordering_l1", "setup_wait_ordering",
        "setup_device_layout", "setup_total", "setup_ordering_cached")
eng = cabi.Engine()
eng.use_hierarchy(H); eng.set_mass(mass)
for rep in range(3):          # rep 0: cold; rep 1, 2: same sparsity pattern -> cached orderings
    t = time.perf_counter(); eng.set_system(lhs); tot = time.perf_counter() - t
    print("set_system %.0f ms:" % (1e3 * tot), {k: round(eng.timing(k)) for k in keys})
    marks = ["pattern_key", "lhs_copied", "upload_U", "upload_A0"] + [f"rap_l{k}" for k in range(1, 6)] + [f"ordering_ready_l{k}" for k in range(6)] + ["device_layout", "tasks_joined", "factor_joined", "mass_done"]
    tl = []
    for m in marks:
        try:
            tl.append((round(eng.timing("t_" + m), 1), m))
        except Exception:
            pass
    print("   timeline:", sorted(tl))
    print("   ordering ms per level:", [round(eng.timing(f"setup_ordering_l{k}"), 1) for k in range(eng.num_levels + 1)], "factor", round(eng.timing("coarsest_solve"), 1))
    t = time.perf_counter(); x, it, res, conv = eng.solve(rhs); print("solve call %.1f ms, cycles %.1f ms, iters %d" % (1e3 * (time.perf_counter() - t), eng.timing("cycles"), it))
