"""Level 0 blocked (gmg_config::block_fine / block_from_level = 0): V-cycles to 1e-4 and in-block share of the couplings with blocks = runs of the
cluster order (default) or grown breadth-first over the operator (GMG_FINE_BLOCKS_GROWN=1), against the colour-major level 0.  One process per setting."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from gravo_mg_amd import cabi
from tests import problems
cases = {"torus48x40": lambda: problems.torus_problem(48, 40, "poisson", 60), "torus96x80": lambda: problems.torus_problem(96, 80, "poisson", 30),
         "cloud3000": lambda: problems.pointcloud_problem(3000), "cloud6000": lambda: problems.pointcloud_problem(6000, lower_bound=100),
         "cloud20000": lambda: problems.pointcloud_problem(20000, lower_bound=200), "cloud120000": lambda: problems.pointcloud_problem(120000, lower_bound=1000)}
for name in sys.argv[1:]:
    P = cases[name]()
    for kw in (dict(block_fine=0, gs_omega=1.0), dict(block_fine=0), dict(block_from_level=0)):
        e = cabi.Engine(**kw); e.set_prolongations(P.U); e.set_mass(P.mass); e.set_system(P.lhs)
        x, it, res, conv = e.solve(P.rhs, tol=1e-6, max_iter=200)
        rec = {"case": name, "kw": kw, "grown": bool(os.environ.get("GMG_FINE_BLOCKS_GROWN")), "iters_1e-6": int(it), "iters_1e-4": int(np.argmax(conv[:, 1] <= 1e-4) + 1), "colors0": e.level_info(0)["n_colors"]}
        if e.level_blocks(0) is not None:
            n2o, _ = e.level_ordering(0); bb, _ = e.level_blocks(0)
            bn = np.repeat(np.arange(len(bb) - 1), np.diff(bb)); blk = np.full(P.lhs.shape[0], -1); real = n2o >= 0; blk[n2o[real]] = bn[real]
            C = sp.coo_matrix(P.lhs); off = C.row != C.col
            rec["in_block_share"] = float(np.mean(blk[C.row[off]] == blk[C.col[off]])); rec["blocks"] = len(bb) - 1; rec["fill"] = float(real.mean())
        print(json.dumps(rec), flush=True)
        e.close()
