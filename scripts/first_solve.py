"""Where the first solve after a set_system spends its time (graph capture / instantiation, staging, cycles)."""
import sys, time
sys.path.insert(0, '.')
import bench
from gravo_mg_amd import cabi
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
for graph in (True, False):
    eng = cabi.Engine(use_graph=graph)
    eng.use_hierarchy(H); eng.set_mass(mass)
    for rep in range(2):
        t = time.perf_counter(); eng.set_system(lhs); ts = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter(); x, it, res, conv = eng.solve(rhs); tv = 1e3 * (time.perf_counter() - t)
        cap = inst = 0.0
        try:
            cap, inst = eng.timing("graph_capture_ms"), eng.timing("graph_instantiate_ms")
        except Exception:
            pass
        print(f"graph={graph} rep={rep}: set_system {ts:.1f} ms, solve call {tv:.1f} ms (cycles {eng.timing('cycles'):.1f}, per-cycle stamps {[round(v, 2) for v in conv[:, 0]]}), "
              f"capture {cap:.2f} ms, instantiate {inst:.2f} ms (cumulative), iters {it}")
    del eng
