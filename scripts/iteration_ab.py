#!/usr/bin/env python3
"""Where does the extra V-cycle come from, and what removes it?  

The reference smoother is lexicographic Gauss-Seidel (multigrid_solver.cpp:1194-1226); the 1-core oracle needs 6 V-cycles to
1e-4 on the 3 M Poisson problem, the multicolour sweep 7.  This script runs the engine variants on the full-size configs and
writes (variant, cycles to 1e-4, residual history, ms per cycle, solve ms):

  python scripts/iteration_ab.py [--configs 4 4r 2 3 1 6] > profiles/r02/iteration_ab.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from gravo_mg_amd import cabi, meshgen  # noqa: E402
import run_configs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["4", "4r", "2", "3", "1", "6"])
    ap.add_argument("--omegas", nargs="+", type=float, default=[1.0, 1.1, 1.15, 1.2, 1.25, 1.3])
    args = ap.parse_args()
    out = []
    for cfg in args.configs:
        name, pos, S, mass, lhs, rhs = run_configs.build(cfg)
        H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S))
        variants = [("exact multicolour GS on every level (block_rows=0), omega=1", dict(block_rows=0, gs_omega=1.0))]
        variants += [(f"default engine, level-0 omega={w:g}", dict(gs_omega=w)) for w in args.omegas]
        for label, kw in variants:
            eng = cabi.Engine(**kw)
            eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
            t = time.perf_counter()
            x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
            solve_ms = 1e3 * (time.perf_counter() - t)
            eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
            t = time.perf_counter(); eng.run_cycles(20, 2); ms_cycle = 1e3 * (time.perf_counter() - t) / 20
            rec = {"config": name, "variant": label, "cycles_to_1e-4": it, "residues": [float(v) for v in conv[:, 1]],
                   "ms_per_cycle": ms_cycle, "solve_call_ms": solve_ms, "cycles_ms": eng.timing("cycles")}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del eng


if __name__ == "__main__":
    main()
