"""Fixed cost of one gmg_run_cycles call (the bench line's K = 20 steps pay it 1/20 each): ms per cycle for K = 1, 5, 20, 200 cycles per call.
  python scripts/call_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gravo_mg_amd import cabi
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
eng = cabi.Engine()
eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
eng.load_problem(rhs, rhs); eng.run_cycles(30, 2)
import torch
for K, calls in ((200, 3), (20, 20), (5, 40), (1, 200), (20, 20), (200, 3)):
    ms = []
    for _ in range(calls):
        torch.cuda.synchronize(); t = time.perf_counter(); eng.run_cycles(K, 2); torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t) / K)
    print(f"K = {K:4d}: median {np.median(ms):.4f} ms per cycle, min {min(ms):.4f}", flush=True)
