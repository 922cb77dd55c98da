"""Per-level kernel times (sweep / residual / restrict / prolong) and the cycle for engine knob settings on a full-size workload.
usage: knob_sweep.py CFG 'k=v,k=v' 'k=v' ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravo_mg_amd import cabi, meshgen
cfg = sys.argv[1]
name, pos, S, mass, lhs, rhs = meshgen.baseline_config(cfg)
H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S))
for v in sys.argv[2:]:
    kw = {k: int(x) for k, x in (a.split('=') for a in v.split(',') if a)}
    eng = cabi.Engine(**kw)
    eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    d = rhs.shape[1]
    line = f"{name[:30]:30s} {v or 'default':28s}"
    for k in range(eng.num_levels):
        ts = [eng.bench_kernel(kind, k, d, 40)[0] for kind in (0, 1, 2, 3)]
        line += f" | L{k} sw {1e3 * ts[0]:5.1f} res {1e3 * ts[1]:5.1f} R {1e3 * ts[2]:5.1f} P {1e3 * ts[3]:5.1f}"
    eng.load_problem(rhs, rhs); eng.run_cycles(3, 2)
    t = time.perf_counter(); eng.run_cycles(30, 2); line += f" | cycle {1e3 * (time.perf_counter() - t) / 30:.3f} ms"
    print(line, flush=True)
    eng.close()
