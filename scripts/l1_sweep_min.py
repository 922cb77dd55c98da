"""A few level-1 block sweeps on the 3 M bench workload and nothing else (target of PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi
import bench as single
H, mass, lhs, rhs = single.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
for d in [int(a) for a in sys.argv[1:]] or [1]:
    t_ms, _ = eng.bench_kernel(0, 1, d, 4)
    print(f"d={d} L1 sweep {1e3 * t_ms:.1f} us")
    t_ms, _ = eng.bench_kernel(0, 0, d, 2)
    print(f"d={d} L0 sweep {1e3 * t_ms:.1f} us")
