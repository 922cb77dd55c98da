"""How many windows of a given span does each 64-row slice of level 0 need (16-bit column codes)?
python scripts/window_stats.py <config>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import bench
from gravo_mg_amd import cabi
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
name, H, mass, lhs, rhs = bench.build_config(cfg)
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
new2old, _ = eng.level_ordering(0)
old2new = np.full(lhs.shape[0], -1, np.int64)
old2new[new2old[new2old >= 0]] = np.nonzero(new2old >= 0)[0]
A = sp.csr_matrix(lhs)
npad = len(new2old)
rng = np.random.default_rng(0)
slices = rng.choice(npad // 64, size=min(3000, npad // 64), replace=False)
for span in (8192, 4096, 2048):
    need = []
    for s in slices:
        rows = new2old[s * 64:(s + 1) * 64]
        rows = rows[rows >= 0]
        cols = []
        for r in rows:
            c = A.indices[A.indptr[r]:A.indptr[r + 1]]
            cols.append(old2new[c[c != r]])
        if not cols:
            need.append(0); continue
        c = np.unique(np.concatenate(cols))
        k, lo = 0, -1
        for v in c:
            if v >= lo:
                k += 1; lo = v + span
        need.append(k)
    need = np.array(need)
    print(name, "span", span, "windows needed: median", int(np.median(need)), "p90", int(np.percentile(need, 90)), "max", int(need.max()),
          "| share of slices with <= 8:", round(float((need <= 8).mean()), 3), "<= 16:", round(float((need <= 16).mean()), 3), "<= 32:", round(float((need <= 32).mean()), 3), flush=True)
