"""Back-substitution time of the coarsest-level solver with 1..8 threads on saved coarsest operators (host only):
python scripts/ldlt_team_bench.py scripts/micro/coarse_4r.npz ...   (GMG_LDLT_BENCH = repetitions, GMG_LDLT_TEAM_PANEL, GMG_LDLT_PARTS)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GMG_LDLT_BENCH", "100")
import numpy as np, scipy.sparse as sp
from gravo_mg_amd import cabi
for path in sys.argv[1:]:
    A = sp.load_npz(path)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    x, nnz = cabi.host_ldlt_solve(A, b)
    print(path, "n", A.shape[0], "nnz(L)", nnz, "residual", np.linalg.norm(A @ x - b) / np.linalg.norm(b), flush=True)
