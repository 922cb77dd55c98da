"""Level-1 block statistics of the 3 M bench workload: entries per block of the explicit / lower parts (what sizes the LDS and register
windows of gs_block_ep), lower entries per row, colours per block; plus the level-1 sweep time at d = 1 and 3."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gravo_mg_amd import cabi
import bench as single
H, mass, lhs, rhs = single.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
pct = [50, 90, 99, 99.9]
for which, name in ((7, "E"), (6, "L")):
    out = eng.debug_sell(1, which)
    ptr = np.asarray(out["slice_ptr"] if "slice_ptr" in out else out["ptr"])
    per_block = ptr[64::64] - ptr[:-64:64]
    per_row = np.diff(ptr)
    print(name, "entries per block: mean %.0f  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d" % (per_block.mean(), *np.percentile(per_block, pct).astype(int), per_block.max()),
          "cap", out.get("row_of", [0])[0], "| per row: mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d" % (per_row.mean(), *np.percentile(per_row, pct).astype(int), per_row.max()))
    for cap in (512, 640, 768, 896, 1024, 1280):
        print("   blocks with more than %d entries: %d of %d" % (cap, int((per_block > cap).sum()), per_block.size))
bb, rc = eng.level_blocks(1)
nb = bb.size - 1
ncol = np.array([rc[bb[i]:bb[i + 1]].max() + 1 for i in range(nb)])
print("blocks", nb, "colours per block: mean %.1f p50 %d p90 %d p99 %d max %d" % (ncol.mean(), *np.percentile(ncol, [50, 90, 99]).astype(int), ncol.max()))
for k in range(eng.num_levels):
    print("level", k, eng.level_info(k))
for d in (1, 3):
    t_ms, _ = eng.bench_kernel(0, 1, d, 50)
    print(f"d={d} L1 sweep {1e3 * t_ms:.1f} us")
