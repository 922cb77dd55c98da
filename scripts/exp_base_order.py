"""EXPERIMENT: random-order 3M torus with different level-0 base orders (cluster order of the hierarchy vs BFS / RCM / two-BFS grid order)."""
import os, sys, time, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import breadth_first_order, reverse_cuthill_mckee, shortest_path
from gravo_mg_amd import meshgen
cfg = sys.argv[1] if len(sys.argv) > 1 else "4r"
name, pos, S, mass, lhs, rhs = meshgen.baseline_config(cfg)
G = sp.csr_matrix(lhs); n = G.shape[0]
t = time.time(); o1, _ = breadth_first_order(G, 0, directed=False); print("bfs", time.time() - t, flush=True)
o1.astype(np.int32).tofile("/tmp/order_bfs.bin")
def bfs_levels(src):
    lev = np.full(n, -1, np.int64); lev[src] = 0; fr = np.array([src]); l = 0
    while fr.size:
        l += 1
        nb = np.unique(np.concatenate([G.indices[G.indptr[v]:G.indptr[v + 1]] for v in fr])) if fr.size < 64 else np.unique(G[fr].indices)
        nb = nb[lev[nb] < 0]; lev[nb] = l; fr = nb
    return lev
t = time.time(); d1 = bfs_levels(0); far = int(np.argmax(d1)); d2 = bfs_levels(int(o1[n // 2]))
print("two level sets", time.time() - t, d1.max(), d2.max(), flush=True)
np.lexsort((np.arange(n), d2, d1)).astype(np.int32).tofile("/tmp/order_grid.bin")
reverse_cuthill_mckee(G, symmetric_mode=True).astype(np.int32).tofile("/tmp/order_rcm.bin")
