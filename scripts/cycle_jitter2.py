"""Per-cycle durations inside ONE solve call (the host's clock at every residual check): are the first cycles of a call slower?
  python scripts/cycle_jitter2.py"""
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
out = np.empty_like(B, order="F")
for rep in range(4):
    x, it, res, conv = eng.solve(B, tol=0.0, max_iter=40, out=out)
    d = np.diff(np.concatenate([[0.0], conv[:, 0]]))
    print("cycle durations (us):", " ".join(f"{1e3 * v:.0f}" for v in d), flush=True)
import torch
for rep in range(3):
    eng.load_problem(B, B); eng.run_cycles(5, 2)
    torch.cuda.synchronize(); t = time.perf_counter(); eng.run_cycles(20, 2); torch.cuda.synchronize(); print("run_cycles(20):", round(1e3 * (time.perf_counter() - t) / 20, 4))
