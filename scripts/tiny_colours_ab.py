"""A/B of gmg_config::merge_tiny_colors (one launch for the tiny leading colour classes of a colour-major level 0) on the full-size workloads
whose colourings leave such classes: ms per V-cycle + check, the level-0 sweep alone, cycles to tolerance.  usage: python scripts/tiny_colours_ab.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gravo_mg_amd import cabi, meshgen

def run(label, H, mass, lhs, rhs, **kw):
    out = {"workload": label}
    for merge in (0, 1):
        e = cabi.Engine(merge_tiny_colors=bool(merge), **kw)
        e.use_hierarchy(H); e.set_mass(mass); e.set_system(lhs)
        e.load_problem(rhs, rhs); e.run_cycles(5, 2)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); e.run_cycles(30, 2); torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t) / 30)
        sweep_ms, launches = e.bench_kernel(0, 0, rhs.shape[1] if rhs.ndim == 2 else 1, 30)
        out["merged" if merge else "launch_per_class"] = {"ms_per_cycle": best, "level0_sweep_ms": sweep_ms, "launches_per_sweep": launches, "colours": e.level_info(0)["n_colors"],
                                                          "tiny_classes": e.timing("tiny_colors_l0"), "tiny_tasks": e.timing("tiny_tasks_l0"), "tiny_rows": e.timing("tiny_rows_l0")}
        e.close()
    print(json.dumps(out), flush=True)

n1 = 1732
Hr, mass_r, lhs_r, rhs_r = bench.build_workload(n1, n1, "random"); run("3M random vertex order", Hr, mass_r, lhs_r, rhs_r); del Hr, lhs_r
Hc, mass_c, lhs_c, rhs_c = bench.build_workload(n1, n1, "chunks"); run("3M shuffled runs of 65 536", Hc, mass_c, lhs_c, rhs_c); del Hc, lhs_c
H, mass, lhs, rhs = bench.build_workload(n1, n1, "natural")
V, F = meshgen.torus_mesh(n1, n1); S, ms = meshgen.cotan_laplacian(V, F)
lhs_b, rhs_b = meshgen.smoothing_system(meshgen.bilaplacian(S, ms), ms, V[:, :1], tau=1e-9)
run("3M Bilaplacian", H, mass, lhs_b, rhs_b); del lhs_b, H
name, pos, S3, mass3, lhs3, rhs3 = meshgen.baseline_config("3")
H3 = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S3), ratio=8.0, lower_bound=1000)
run("2M point cloud, colour-major level 0", H3, mass3, lhs3, rhs3, block_fine=0)
