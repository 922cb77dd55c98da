"""Restriction time and SELL padding against the length-sorting window of U^T, on four full-size workloads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi, meshgen
for cfg in ("4", "4r", "3", "6"):
    name, pos, S, mass, lhs, rhs = meshgen.baseline_config(cfg)
    H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S))
    for sg in (0, 64, 128, 256, 1024):
        eng = cabi.Engine(restrict_sigma=sg)
        eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        line = f"{name[:44]:44s} restrict_sigma={sg:5d}"
        for k in (0, 1):
            t_ms, _ = eng.bench_kernel(2, k, 1, 50)
            info = eng.debug_sell(k, 4)
            line += f" | L{k} restrict {1e3 * t_ms:5.1f} us, stored/real {len(info['val']) / max((info['val'] != 0).sum(), 1):.3f}"
        eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
        t = time.perf_counter(); eng.run_cycles(20, 2); line += f" | cycle {50 * (time.perf_counter() - t):.3f} ms"
        print(line, flush=True)
        eng.close()
