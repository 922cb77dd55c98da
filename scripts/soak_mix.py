"""Second soak: what several threads of one process may do at once -- build hierarchies, set systems on engines of different kinds (default, fp32
inner cycle, device coarse apply, hipGraph replay, host planner), solve n x 3 blocks, destroy -- each thread's results must repeat its own
single-threaded run bit for bit."""
import sys, time, threading, os, faulthandler
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp
from gravo_mg_amd import cabi, meshgen
WD = int(os.environ.get('GMG_SOAK_WATCHDOG', '0'))
def mesh(n1, n2):
    V, F = meshgen.torus_mesh(n1, n2); S, mass = meshgen.cotan_laplacian(V, F)
    return V, S, mass, meshgen.neighbors_from_stiffness(S)
M = [mesh(120, 100), mesh(96, 80), mesh(150, 140)]
P = meshgen.torus_points(15000, noise=0.002); Sp, mp = meshgen.knn_graph_laplacian(P, 8); M.append((P, Sp, mp, meshgen.neighbors_from_stiffness(Sp)))
KW = [dict(), dict(inner_precision=1), dict(coarse_mode=cabi.COARSE_DEVICE_INVERSE), dict(use_graph=True)]
def run(i, out):
    V, S, mass, neigh = M[i]
    sig = []
    for rep in range(2):
        H = cabi.Hierarchy(V, neigh, lower_bound=300)
        eng = cabi.Engine(**KW[i]); eng.use_hierarchy(H); eng.set_mass(mass)
        for tau in (1e-3, 3e-3):
            lhs = sp.csc_matrix(sp.diags(mass) + tau * S)
            eng.set_system(lhs)
            rhs = np.asfortranarray(mass[:, None] * V[:, :3]) if V.shape[1] >= 3 else np.asfortranarray(mass[:, None] * V)
            x, it, res, _ = eng.solve(rhs, tol=1e-6, max_iter=100)
            sig.append((int(it), float(res), float(np.abs(x).sum())))
        eng.close(); del H
    out.append(sig)
t0 = time.time(); ref = []
for i in range(len(M)): run(i, ref)
print("sequential pass %.1f s" % (time.time() - t0), flush=True)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    outs = [[] for _ in M]
    if WD: faulthandler.dump_traceback_later(WD, exit=True)
    th = [threading.Thread(target=run, args=(i, outs[i])) for i in range(len(M))]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(len(M)): assert outs[i][0] == ref[i], (rnd, i, outs[i][0][:2], ref[i][:2])
    if rnd % 10 == 0: print("concurrent round", rnd, "ok %.1f s" % (time.time() - t0), flush=True)
print("soak ok")
