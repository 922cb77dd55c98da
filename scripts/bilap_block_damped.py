"""Bilaplacian (config 5b, M + 1e-9 S M^-1 S, 3 M vertices): level 0 blocked with an UNDER-relaxed block sweep (fine_block_omega < 1) against the colour-major level 0 --
the plain block sweep (omega = 1) diverges on this operator (not a Stieltjes matrix).  Residues after 10 / 20 / 40 cycles, ms per cycle."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravo_mg_amd import cabi, meshgen
name, pos, S, mass, lhs, rhs = meshgen.baseline_config("5b")
H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
for kw in ({}, {"gs_omega": 1.0}, {"block_from_level": 0, "fine_block_omega": 0.9}, {"block_from_level": 0, "fine_block_omega": 0.8}, {"block_from_level": 0, "fine_block_omega": 0.7}, {"block_from_level": 0, "fine_block_omega": 0.6}):
    eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    eng.load_problem(rhs, rhs)
    t = time.perf_counter(); hist = eng.run_cycles(40, 2); ms = 1e3 * (time.perf_counter() - t) / 40
    print(json.dumps({"kw": kw, "ms_per_cycle": round(ms, 4), "residue_after": {k: float("%.3g" % hist[k - 1]) for k in (1, 5, 10, 20, 40)}}), flush=True)
    eng.close()
