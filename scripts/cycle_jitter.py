"""Per-cycle wall time inside gmg_solve over many cycles of the bench workload: median, 99th percentile, maximum (are there stalls?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as single
from gravo_mg_amd import cabi
H, mass, lhs, rhs = single.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
all_dt = []
for rep in range(10):
    x, it, res, conv = eng.solve(rhs, tol=0.0, max_iter=100)
    t = conv[:, 0]
    all_dt += list(np.diff(t))
dt = 1e3 * np.array(all_dt)
print(f"{len(dt)} cycles: median {np.median(dt):.1f} us, p90 {np.percentile(dt, 90):.1f}, p99 {np.percentile(dt, 99):.1f}, max {dt.max():.1f} us; cycles above 1.5x median: {int((dt > 1.5 * np.median(dt)).sum())}")
