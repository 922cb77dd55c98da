"""Marks (ms since entry) of a COLD gmg_set_system -- no structure prepared -- at the bench workload, on a process whose pools and code objects are warm.
usage: python scripts/cold_setup_marks.py [natural|random] [partition_world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gravo_mg_amd import cabi
order = sys.argv[1] if len(sys.argv) > 1 else "natural"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, mass, lhs, rhs = bench.build_workload(1732, 1732, order)
def run():
    eng = cabi.Engine(prepare_structure=0, **({"row_align": 64 * world, "block_fine": 0} if world > 1 else {}))
    if world > 1:
        eng.dist_partition(0, world)
    eng.use_hierarchy(H); eng.set_mass(mass)
    t = time.perf_counter(); eng.set_system(lhs); ms = 1e3 * (time.perf_counter() - t)
    marks = {}
    for m in ["pattern_key", "permuted_pattern", "upload_A0"] + [f"rap_l{k}" for k in range(1, 6)] + [f"ordering_ready_l{k}" for k in range(6)] + ["coarse_inverse_early", "device_layout", "tasks_joined", "factor_joined", "mass_done"]:
        try:
            marks[m] = round(eng.timing("t_" + m), 2)
        except Exception:
            pass
    marks["colored_ahead"] = eng.timing("setup_colored_ahead")
    for k in range(5):
        try:
            marks[f"(ordering task of level {k}: ms)"] = round(eng.timing(f"setup_ordering_l{k}"), 2)
        except Exception:
            pass
    return ms, sorted(marks.items(), key=lambda kv: kv[1])
run()
for _ in range(2):
    ms, marks = run()
    print("COLD set_system %.2f ms" % ms, marks, flush=True)
