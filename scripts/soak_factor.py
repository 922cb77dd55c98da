"""Soak of the threaded set-up paths: several handles re-setting systems of the live pattern (new tau: values-only refresh, threaded LDL^T) and of a
new pattern (cold set-up) from several Python threads at once (ctypes releases the GIL: the handles' factorisation teams, the worker pool and the
polling threads compete); every solve must repeat the single-threaded run's result bit for bit."""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from gravo_mg_amd import cabi, meshgen
from tests import problems

def systems(P, taus):
    S, mass = P.S, P.mass
    import scipy.sparse as sp
    return [(sp.csc_matrix(sp.diags(mass) * t + S), P.rhs) for t in taus]

cases = [problems.torus_problem(200, 180, "poisson", 1000), problems.pointcloud_problem(20000, lower_bound=1000), problems.torus_problem(96, 80, "poisson", 200)]
taus = [1e-3, 2e-3, 5e-4, 1e-2, 1e-3]
def run(P, reps, out):
    eng = cabi.Engine(); eng.set_prolongations(P.U); eng.set_mass(P.mass)
    sig = []
    for r in range(reps):
        for lhs, rhs in systems(P, taus):
            eng.set_system(lhs)
            x, it, res, _ = eng.solve(rhs, tol=1e-6, max_iter=100)
            sig.append((int(it), float(res), float(np.abs(x).sum())))
        if r % 2 == 1:      # a different pattern in between: cold set-up, orderings rebuilt
            import scipy.sparse as sp
            eng.set_system(sp.csc_matrix(sp.diags(P.mass) + 1e-3 * P.S + 1e-9 * (P.S @ P.S)))
    out.append(sig); eng.close()
import os, faulthandler
WD = int(os.environ.get('GMG_SOAK_WATCHDOG', '0'))
t0 = time.time()
ref = []
for P in cases: run(P, 2, ref)
print("sequential pass %.1f s" % (time.time() - t0), [len(s) for s in ref], flush=True)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    outs = [[] for _ in cases]
    if WD: faulthandler.dump_traceback_later(WD, exit=True)
    th = [threading.Thread(target=run, args=(P, 2, outs[i])) for i, P in enumerate(cases)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(len(cases)):
        assert outs[i][0] == ref[i], (rnd, i, [a for a, b in zip(outs[i][0], ref[i]) if a != b][:3])
    print("concurrent round", rnd, "ok %.1f s" % (time.time() - t0), flush=True)
print("soak ok")
