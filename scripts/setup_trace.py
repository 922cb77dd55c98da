"""One cold gmg_set_system at the bench workload (and one same-pattern repeat), for `rocprofv3 --kernel-trace --memory-copy-trace`:
scripts/setup_timeline.py turns the two traces into a timeline of the call (kernels, copies, idle gaps).
usage: python scripts/setup_trace.py [natural|random] [n1]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gravo_mg_amd import cabi
order = sys.argv[1] if len(sys.argv) > 1 else "natural"
N1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1732
H, mass, lhs, rhs = bench.build_workload(N1, N1, order)
eng = cabi.Engine()
eng.use_hierarchy(H); eng.set_mass(mass)
warm = cabi.Engine(); warm.use_hierarchy(H); warm.set_mass(mass); warm.set_system(lhs); del warm      # code objects loaded, pools of the process warm
time.sleep(0.2)
t = time.perf_counter(); eng.set_system(lhs); tot = time.perf_counter() - t
print("SETUP_TRACE cold set_system %.2f ms" % (1e3 * tot), flush=True)
marks = {}
print("SETUP_TRACE prepared", eng.timing("setup_structure_prepared"), "structure_prepare_ms", round(eng.timing("structure_prepare_ms"), 2) if eng.timing("setup_structure_prepared") else None, flush=True)
for m in ["pattern_key", "upload_A0"] + [f"rap_l{k}" for k in range(1, 6)] + [f"ordering_ready_l{k}" for k in range(6)] + ["device_layout", "tasks_joined", "factor_joined", "mass_done"]:
    try:
        marks[m] = round(eng.timing("t_" + m), 2)
    except Exception:
        pass
print("SETUP_TRACE marks", sorted(marks.items(), key=lambda kv: kv[1]), flush=True)
x, it, res, conv = eng.solve(rhs)
x, it, res, conv = eng.solve(rhs)          # (the second solve: staging and vectors exist)
print("SETUP_TRACE load keys", {k: round(eng.timing(k), 3) for k in ("load_vectors", "load_b", "load_x", "load_sync")})
print("SETUP_TRACE solve iters", it, "solve_call", round(eng.timing("solve_call"), 2), "load", round(eng.timing("solve_load"), 2), "fetch", round(eng.timing("solve_fetch"), 2), "cycles", round(eng.timing("cycles"), 2))
