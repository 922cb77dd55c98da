"""Set-up cost and cycle time of the device coarse solve against the host one, at the coarsest sizes of the bench workloads.
  python scripts/coarse_inverse_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gravo_mg_amd import cabi, meshgen

def run(tag, n1, n2, kind, d):
    V, F = meshgen.torus_mesh(n1, n2)
    S, mass = meshgen.cotan_laplacian(V, F)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=1000)
    lhs, rhs = (meshgen.smoothing_system(S, mass, V) if kind == "smoothing" else meshgen.poisson_system(S, mass, d=d))
    for mode, name in ((cabi.COARSE_AUTO, "device"), (cabi.COARSE_HOST_LDLT, "host")):
        eng = cabi.Engine(coarse_mode=mode)
        eng.use_hierarchy(H); eng.set_mass(mass)
        t = time.perf_counter(); eng.set_system(lhs); t1 = 1e3 * (time.perf_counter() - t)
        lhs2 = lhs.copy(); lhs2.data *= 1.001
        t = time.perf_counter(); eng.set_system(lhs2); t2 = 1e3 * (time.perf_counter() - t)
        eng.load_problem(rhs, rhs); eng.run_cycles(12, 2)      # (host placement: the helper team and the factor warm up over the first cycles)
        t = time.perf_counter(); eng.run_cycles(50, 2); cyc = 1e3 * (time.perf_counter() - t) / 50
        legs = eng.profile_cycle(2, 10)
        L = eng.num_levels
        keys = {k: round(eng.timing(k), 3) for k in ("coarse_inverse_ms", "coarse_inverse_export_ms", "coarse_inverse_levels", "coarse_inverse_chunks", "coarse_inverse_upload_ms", "coarse_inverse_tiles_ms", "coarsest_solve") if _has(eng, k)}
        print(f"{tag} {name}: n_L={eng.level_info(L)['n']} set_system {t1:.2f} / again {t2:.2f} ms, cycle {cyc:.4f} ms, legs {[round(float(v), 4) for v in legs]} {keys}", flush=True)
        eng.close()

def _has(eng, k):
    try:
        eng.timing(k); return True
    except Exception:
        return False

if __name__ == "__main__":
    run("36k d3", 190, 190, "smoothing", 3)
    run("722k d1", 850, 850, "poisson", 1)
    run("3M d1", 1732, 1732, "poisson", 1)
    run("3M d3", 1732, 1732, "smoothing", 3)
