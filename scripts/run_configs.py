#!/usr/bin/env python3
"""BASELINE.json's configs at FULL size on one MI355X (SURVEY.md 8d's synthetic stand-ins), with the
size-independent checks the domain offers: the solve reaches the reference's stopping test, the oracle's
residualCheck of the returned solution agrees with the GPU's, the residual history contracts monotonically, and
(where it finishes in seconds) the 1-core oracle needs the same number of V-cycles.

  python scripts/run_configs.py [--configs 1 2 3 4 4r] [--oracle-max-n 800000] > profiles/r01/configs.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gravo_mg_amd import cabi, meshgen  # noqa: E402


def build(cfg):
    return meshgen.baseline_config(cfg)


def run_bilaplacian(name, H, mass, lhs, rhs, t_build, t_hier):
    """Config 5: the reference V-cycle contracts slowly on 4th-order operators (factor ~0.9+ per cycle), so
    parity is 'same residual history as the fp64 path / the CPU restatement after N cycles' plus time per cycle."""
    from oracle import oracle
    rec = {"config": name, "n": int(lhs.shape[0]), "nnz": int(lhs.nnz), "d": int(rhs.shape[1]), "checks": {}}
    hist = {}
    for label, prec in (("fp64", 0), ("mixed", 1)):
        eng = cabi.Engine(inner_precision=prec)
        eng.use_hierarchy(H); eng.set_mass(mass)
        t = time.perf_counter(); eng.set_system(lhs); rec[f"set_system_ms_{label}"] = 1e3 * (time.perf_counter() - t)
        eng.load_problem(rhs, rhs)
        hist[label] = eng.run_cycles(12, 2)
        eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
        t = time.perf_counter(); eng.run_cycles(10, 2); rec[f"ms_per_cycle_{label}"] = 1e2 * (time.perf_counter() - t)
        if prec == 1:
            x = eng.fetch_solution()
            rec["checks"]["oracle_residual_check_of_mixed_iterate"] = oracle.residual_check(lhs, mass, rhs, x, 2)
        rec["levels"] = [eng.level_info(k)["n"] for k in range(eng.num_levels + 1)]
        del eng
    rec["history_fp64"] = [float(v) for v in hist["fp64"]]
    rec["history_mixed"] = [float(v) for v in hist["mixed"]]
    rec["checks"]["mixed_history_matches_fp64"] = bool(np.allclose(hist["mixed"], hist["fp64"], rtol=2e-3))
    rec["history_contracts"] = bool(np.all(np.diff(hist["mixed"]) < 0))     # informational: see the oracle's own history
    O = oracle.Hierarchy(H.U, mass)
    t = time.perf_counter(); O.set_system(lhs)
    xo, ito, reso, convo = O.solve(rhs, tol=0.0, max_iter=3)
    rec["oracle"] = {"history": [float(v) for v in convo[:, 1]], "total_s": time.perf_counter() - t, "ms_per_cycle": float(convo[-1, 0] / ito)}
    # lexicographic (oracle) vs multicolour (GPU) Gauss-Seidel: same smoother class, different sweep order.  With the
    # reference's tau = 1e-3 the reference algorithm itself does not contract on this mesh (the oracle's residual GROWS
    # from x0 = rhs); parity is "same behaviour": same trend, same magnitude within a factor 2.
    ratio = hist["fp64"][:3] / convo[:, 1]
    rec["checks"]["oracle_same_magnitude"] = bool(np.all((ratio > 0.5) & (ratio < 2.0)))
    rec["checks"]["oracle_same_trend"] = bool(np.sign(hist["fp64"][2] - hist["fp64"][0]) == np.sign(convo[2, 1] - convo[0, 1]))
    rec["hierarchy_s"], rec["input_build_s"] = t_hier, t_build
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["1", "2", "3", "4", "4r"])
    ap.add_argument("--oracle-max-n", type=int, default=800_000, help="run the full CPU oracle solve up to this size")
    args = ap.parse_args()
    out = []
    for cfg in args.configs:
        t = time.perf_counter()
        name, pos, S, mass, lhs, rhs = build(cfg)
        t_build = time.perf_counter() - t
        neigh = meshgen.neighbors_from_stiffness(S)
        t = time.perf_counter()
        H = cabi.Hierarchy(pos, neigh)
        t_hier = time.perf_counter() - t
        if cfg in ("5", "5b"):
            rec = run_bilaplacian(name, H, mass, lhs, rhs, t_build, t_hier)
            print(json.dumps(rec), flush=True)
            out.append(rec)
            continue
        eng = cabi.Engine()
        eng.use_hierarchy(H)
        eng.set_mass(mass)
        t = time.perf_counter()
        eng.set_system(lhs)
        t_setup = time.perf_counter() - t
        t = time.perf_counter()
        x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
        t_solve = time.perf_counter() - t
        eng.load_problem(rhs, rhs)
        eng.run_cycles(3, 2)
        t = time.perf_counter()
        eng.run_cycles(10, 2)
        ms_cycle = 1e2 * (time.perf_counter() - t)
        rec = {
            "config": name, "n": int(lhs.shape[0]), "nnz": int(lhs.nnz), "d": int(rhs.shape[1]),
            "levels": [eng.level_info(k)["n"] for k in range(eng.num_levels + 1)],
            "colors": [eng.level_info(k)["n_colors"] for k in range(eng.num_levels + 1)],
            "gpu_iterations": it, "gpu_residue": res, "residue_history": [float(v) for v in conv[:, 1]],
            "ms_per_cycle": ms_cycle, "set_system_ms": 1e3 * t_setup, "solve_call_ms": 1e3 * t_solve,
            "hierarchy_s": t_hier, "input_build_s": t_build,
            "checks": {},
        }
        from oracle import oracle
        chk = oracle.residual_check(lhs, mass, rhs, x, 2)
        rec["checks"]["reaches_tolerance"] = bool(res <= 1e-4 and it < 100)
        rec["checks"]["oracle_residual_check_of_gpu_solution"] = chk
        rec["checks"]["oracle_agrees"] = bool(abs(chk - res) <= 1e-3 * res + 1e-7)
        rec["checks"]["history_contracts"] = bool(np.all(np.diff(conv[:, 1]) < 0))
        if lhs.shape[0] <= args.oracle_max_n:
            O = oracle.Hierarchy(H.U, mass)
            t = time.perf_counter()
            O.set_system(lhs)
            xo, ito, reso, convo = O.solve(rhs, tol=1e-4)
            rec["oracle"] = {"iterations": ito, "residue": reso, "total_s": time.perf_counter() - t,
                             "ms_per_cycle": float(convo[-1, 0] / ito)}
            m = mass[:, None]
            rec["oracle"]["history"] = [float(v) for v in convo[:, 1]]
            rec["iterations_gpu_vs_oracle"] = [int(it), int(ito)]
            rec["checks"]["iterations_equal_oracle"] = bool(it == ito)          # informational (different sweep order): not part of all_checks_pass
            rec["checks"]["iterations_not_more_than_oracle_plus_1"] = bool(it <= ito + 1)
            rec["checks"]["solution_distance_M"] = float(np.sqrt((m * (x - xo) ** 2).sum() / (m * xo ** 2).sum()))
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del eng
    ok = all(all(v for k, v in r["checks"].items() if isinstance(v, bool) and k != "iterations_equal_oracle") for r in out)
    print(json.dumps({"all_checks_pass": ok}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
