import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gravo_mg_amd import cabi
import bench as single
H, mass, lhs, rhs = single.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
for which, name in ((7, "E"), (6, "L")):
    out = eng.debug_sell(1, which)
    ptr = out["slice_ptr"] if "slice_ptr" in out else out["ptr"]
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    ptr = np.asarray(ptr)
    per_block = ptr[64::64] - ptr[:-64:64]
    print(name, "entries per block: mean %.0f  p50 %d  p90 %d  p99 %d  max %d" % (per_block.mean(), *np.percentile(per_block, [50, 90, 99]).astype(int), per_block.max()), "cap", out.get("row_of", [0])[0])
