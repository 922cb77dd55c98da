"""Residual histories of a config with level 0 as a colour-major level (default) and as a blocked level (block_from_level=0), with the oracle's check of the final iterate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
from oracle import oracle
cfg = sys.argv[1]
name, pos, S, mass, lhs, rhs = meshgen.baseline_config(cfg)
H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
for kw in ({}, {"block_from_level": 0}, {"gs_omega": 1.0}):
    eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass)
    t = time.perf_counter(); eng.set_system(lhs); ts = 1e3 * (time.perf_counter() - t)
    x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=60)
    chk = oracle.residual_check(lhs, mass, rhs, x, 2)
    print(cfg, kw, "set_system %.0f ms" % ts, "iters", it, "residue", res, "oracle check", chk, "history", ["%.2e" % v for v in conv[:, 1][:24]], flush=True)
    for k in ("setup_ordering", "setup_ordering_l0", "setup_device_layout", "reduction", "upload"):
        print("   ", k, round(eng.timing(k), 1), end="")
    print()
    eng.close()
