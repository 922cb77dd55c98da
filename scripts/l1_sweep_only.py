"""Only level-k sweeps of the bench workload, for counter collection: python scripts/l1_sweep_only.py [k] [d] [key=value ...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
from gravo_mg_amd import cabi
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kw = {a.split('=')[0]: int(a.split('=')[1]) for a in sys.argv[3:]}
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
eng = cabi.Engine(**kw); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
b = np.repeat(rhs, d, axis=1)
eng.load_problem(b, b)
ms, launches = eng.bench_kernel(0, k, d, 20)
print("sweep ms", ms, "launches", launches)
