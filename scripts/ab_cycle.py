"""A/B aid: ms per V-cycle (+ residual check) of one BASELINE config, best and median of --reps timed batches of --steps cycles in ONE
process (set-up once).  Environment knobs (GMG_*) are read by the library at first use: run once per setting.
  python scripts/ab_cycle.py --config 4r [--steps 100] [--reps 5] [--label text] [engine options as key=value ...]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="4"); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--label", default=""); ap.add_argument("opts", nargs="*")
a = ap.parse_args()
import torch
from gravo_mg_amd import cabi, meshgen
name, pos, S, mass, lhs, rhs = meshgen.baseline_config(a.config)
H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
kw = {}
for o in a.opts:
    k, v = o.split("="); kw[k] = float(v) if "." in v else int(v)
eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass)
t = time.perf_counter(); eng.set_system(lhs); set_ms = 1e3 * (time.perf_counter() - t)
x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
ts = []
for r in range(a.reps):
    c0 = eng.timing("coarse_host_ms")
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run_cycles(a.steps, 2); torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0) / a.steps)
    coarse_us = 1e3 * (eng.timing("coarse_host_ms") - c0) / a.steps
print(json.dumps({"label": a.label, "config": a.config, "best_ms": min(ts), "median_ms": float(np.median(ts)), "all_ms": [round(v, 4) for v in ts], "iterations": int(it),
                  "set_system_ms": set_ms, "levels": [eng.level_info(k)["n"] for k in range(eng.num_levels + 1)], "colors": [eng.level_info(k)["n_colors"] for k in range(eng.num_levels + 1)],
                  "coarse_host_us_per_cycle": coarse_us}))
