"""Soak test of the polled completion path: many solves / cycles on several handles, results must repeat exactly."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from gravo_mg_amd import cabi
from tests import problems

t0 = time.time()
Ps = [problems.torus_problem(96, 80, "poisson", 30), problems.torus_problem(64, 60, "smoothing", 60), problems.torus_problem(200, 180, "poisson", 200)]
engs = []
for P in Ps:
    for coarse in (cabi.COARSE_HOST_LDLT, cabi.COARSE_DEVICE_INVERSE):
        e = cabi.Engine(coarse_mode=coarse)
        e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
        engs.append((e, P))
ref = {}
n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
for it in range(n_solves):
    k = int(rng.integers(len(engs)))
    e, P = engs[k]
    x, iters, res, conv = e.solve(P.rhs, tol=1e-6, stop_type=int(rng.integers(0, 3)) if False else 2, max_iter=60)
    key = k
    sig = (iters, float(res), float(np.abs(x).sum()))
    if key in ref:
        assert ref[key] == sig, (it, k, ref[key], sig)
    else:
        ref[key] = sig
    if it % 50 == 0:
        # a long run of cycles with a check after each: the sequence words must never be missed or read early
        e.load_problem(P.rhs, P.rhs)
        r = e.run_cycles(200, 2)
        assert np.all(np.isfinite(r)) and r[-1] <= r[0]
        r2 = None
        e.load_problem(P.rhs, P.rhs)
        r2 = e.run_cycles(200, 2)
        assert np.array_equal(r, r2), "residual history not reproducible"
print(f"soak ok: {n_solves} solves on {len(engs)} handles in {time.time() - t0:.1f}s; signatures {ref}")
