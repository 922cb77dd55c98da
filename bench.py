#!/usr/bin/env python3
"""bench.py -- the Gravo MG V-cycle hot path on MI355X: V-cycle wall time + solve-to-1e-4 iterations on a
~3 M-vertex mesh Poisson problem (BASELINE.json metric), with the fine-level kernel's HBM roofline and the
1-core CPU restatement timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n1 1732 --n2 1732]

A "step" is ONE V-cycle followed by the residual check, exactly one trip of the reference's solve loop
(gravomg/src/multigrid_solver.cpp:1411-1417), on device-resident data.  Workload:
jittered torus 1732 x 1732 = 2 999 824 vertices, cotangent Laplacian, lhs = 1e-6*M + S, rhs = M*y,
y ~ N(0,1) seed 42, d = 1, x0 = rhs, ratio 8, lower_bound 1000, 2+2 smoothing sweeps, M-norm stop at 1e-4.

For N > 1 the driver launches this file under torch.distributed.run; the finest level is row-partitioned and after
every colour sweep each rank stores the halo entries its peers read straight into their mailboxes (device-initiated
peer-to-peer exchange, gravo_mg_amd/csrc/engine_dist.hip.hpp; RCCL all-gathers as the fallback, gravo_mg_amd/dist.py).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(n1, n2, order="natural"):
    from gravo_mg_amd import cabi, meshgen
    t = time.perf_counter()
    V, F = meshgen.torus_mesh(n1, n2, order=order)
    S, mass = meshgen.cotan_laplacian(V, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    log(f"[bench] mesh + Laplacian: n={V.shape[0]} nnz={S.nnz} ({time.perf_counter() - t:.1f}s)")
    t = time.perf_counter()
    H = cabi.Hierarchy(V, neigh, ratio=8.0, lower_bound=1000)
    log(f"[bench] hierarchy: dof={[H.U[0].shape[0]] + [u.shape[1] for u in H.U]} ({time.perf_counter() - t:.1f}s)")
    lhs, rhs = meshgen.poisson_system(S, mass, tau=1e-6, seed=42, d=1)
    return H, mass, lhs, rhs


def build_config(cfg):
    """Another BASELINE config as the main workload (profiling aid; the driver's line is the default workload)."""
    from gravo_mg_amd import cabi, meshgen
    name, pos, S, mass, lhs, rhs = meshgen.baseline_config(cfg)
    H = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S), ratio=8.0, lower_bound=1000)
    log(f"[bench] {name}: n={lhs.shape[0]} nnz={lhs.nnz} d={rhs.shape[1]}")
    return name, H, mass, lhs, rhs


def load_pmc_traffic(workload):
    """HBM bytes per fine-level launch from the committed rocprofv3 --pmc summary (profiles/), if one matches: (value, source).
    The counters cannot be read inside this run (rocprofv3 wraps the process), so the figure is STATIC: measured once per
    build in a separate --pmc pass of this same command and committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
        if j.get("workload") == workload:
            return j.get("hbm_bytes_per_launch"), "static: profiles/pmc_traffic.json (" + j.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes") + ")"
    except Exception:
        pass
    return None, None


def variant_run(cabi, torch, label, H, mass, lhs, rhs, steps, warmup, kernels=False, levels=False, reach=None, **kw):
    """ms per V-cycle (incl. residual check) + solve-to-1e-4 of another workload / engine variant (never `value`).
    kernels: also the fine-level kernels one by one (HIP events) with their algorithmic GB/s for this right-hand-side width."""
    import numpy as np
    eng = cabi.Engine(**kw)
    eng.use_hierarchy(H); eng.set_mass(mass)
    t = time.perf_counter(); eng.set_system(lhs); set_ms = 1e3 * (time.perf_counter() - t)
    # the C-ABI takes column-major n x d blocks (Eigen::MatrixXd, what a reference caller holds): converted once, outside the timed calls; the
    # second solve writes into the first one's result array (round-4 verdict: the 5 ms between second_solve_ms and solve_call at d = 3 were the
    # row-major -> column-major copy of the right-hand side and the first touch of a fresh 72 MB result, both inside the timed statement)
    rhs = np.asfortranarray(rhs)
    t = time.perf_counter(); x, it, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100); solve_ms = 1e3 * (time.perf_counter() - t)
    first = {k: eng.timing(k) for k in ("solve_load", "cycles", "solve_fetch", "solve_call")}
    xbuf = x if (isinstance(x, np.ndarray) and x.ndim == 2 and x.flags.f_contiguous) else None
    t = time.perf_counter(); eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100, out=xbuf); second_ms = 1e3 * (time.perf_counter() - t)
    second = {k: eng.timing(k) for k in ("solve_load", "cycles", "solve_fetch", "solve_call")}
    eng.load_problem(rhs, rhs); eng.run_cycles(warmup, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run_cycles(steps, 2); torch.cuda.synchronize()
    out = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / steps, "iterations_to_1e-4": int(it), "residues": [float(v) for v in conv[:, 1]],
           "coarse_on_device": bool(eng.timing("coarse_on_device")), "coarse_inverse_ms": (eng.timing("coarse_inverse_ms") if eng.timing("coarse_on_device") else None),
           "solve_ms": solve_ms, "first_solve_ms": solve_ms, "second_solve_ms": second_ms, "first_solve_timing_ms": first, "second_solve_timing_ms": second,
           "set_system_ms": set_ms, "n_vertices": int(lhs.shape[0]),
           "levels": [eng.level_info(k)["n"] for k in range(eng.num_levels + 1)], "colors": [eng.level_info(k)["n_colors"] for k in range(eng.num_levels + 1)],
           "level0_sweep": "multicolour (one launch per colour)" if eng.level_blocks(0) is None else
                           "block-hybrid (one launch per sweep; chosen by gmg_config::block_fine: long rows, Stieltjes signs)"}
    d = int(rhs.shape[1])
    if reach is not None:      # systems the reference iteration itself does not bring to 1e-4 within max_iter: cycles to a looser mark
        out["reach"] = {"residue": reach, "cycles": int(next((i + 1 for i, r in enumerate(conv[:, 1]) if r <= reach), -1)), "last_residue": float(conv[-1, 1])}
        if len(out["residues"]) > 12:
            out["residues"] = out["residues"][:6] + out["residues"][-6:]
    mixed = bool(kw.get("inner_precision"))
    cyc_bytes = mixed_cycle_bytes(eng, H, d) if mixed else cycle_algorithmic_bytes(eng, d)
    if mixed:
        out["byte_model"] = "fp32 inner cycle (s = 4) + fp64 defect / norm pass + fp64 correction"
    if levels and not mixed:
        out["levels_roofline"] = level_roofline(eng, rhs)
    out["cycle_algorithmic_GB"] = cyc_bytes / 1e9
    out["cycle_frac_of_peak"] = cyc_bytes / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if kernels:
        kk = {}
        for name, kind, k in (("fine_sweep", 0, 0), ("fine_residual", 1, 0), ("fine_restrict", 2, 0), ("fine_prolong_add", 3, 0), ("fine_norm", 4, 0),
                              ("level1_sweep", 0, 1), ("level1_residual", 1, 1)):
            if k >= eng.num_levels:
                continue
            ms, launches = eng.bench_kernel(kind, k, d, 30)
            by = eng.algorithmic_bytes(kind, k, d)
            kk[name] = {"ms": ms, "launches": launches, "algorithmic_MB": by / 1e6, "GBps": by / (ms * 1e-3) / 1e9}
        out["kernels"] = kk
    eng.close()
    log(f"[bench] variant {label}: {out['ms_per_step']:.3f} ms/cycle, {it} cycles to 1e-4")
    return out


def cpu_quota():
    """CPUs this container may use (cgroup v2 / v1 quota), or None when unlimited -- os.cpu_count() reports the host's."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cycle_algorithmic_bytes(eng, d):
    """one V-cycle incl. residual check = sum over the smoothed levels of (pre + post sweeps + residual) sweeps +
    restriction + prolongation, + the check on level 0 -- each array counted once per kernel."""
    total = 0.0
    for k in range(eng.num_levels):
        total += (eng.pre_iters + eng.post_iters + 1) * eng.algorithmic_bytes(0, k, d) + eng.algorithmic_bytes(2, k, d) + eng.algorithmic_bytes(3, k, d)
    return total + eng.algorithmic_bytes(4, 0, d)


def level_roofline(eng, rhs, reps=10):
    """Where a cycle's time goes, level by level (HIP events at the leg boundaries, gmg_profile_cycle), against the algorithmic bytes of
    SURVEY.md 8(d) for that level: [(pre + post + 1) sweep-equivalents + restriction + prolongation] of level k, the coarsest solve with its
    host round trip, the residual check."""
    d = int(rhs.shape[1])
    eng.load_problem(rhs, rhs); eng.run_cycles(2, 2)
    legs = eng.profile_cycle(2, reps)
    L = eng.num_levels
    out = []
    for k in range(L):
        by = (eng.pre_iters + eng.post_iters + 1) * eng.algorithmic_bytes(0, k, d) + eng.algorithmic_bytes(2, k, d) + eng.algorithmic_bytes(3, k, d)
        out.append({"level": k, "rows": eng.level_info(k)["n"], "ms": float(legs[k]), "algorithmic_bytes": by, "GBps": by / (legs[k] * 1e-3) / 1e9,
                    "frac": by / (legs[k] * 1e-3) / 1e9 / HBM_PEAK_GBS})
    by = eng.algorithmic_bytes(4, 0, d)
    nl = eng.level_info(L)["n"]
    if eng.timing("coarse_on_device"):
        by_c = 8.0 * nl * nl + 4.0 * nl + 2 * 8.0 * nl * d
        out.append({"level": "coarsest solve (dense inverse applied on the device, %d unknowns)" % nl, "ms": float(legs[L]), "algorithmic_bytes": by_c,
                    "GBps": by_c / (legs[L] * 1e-3) / 1e9, "frac": by_c / (legs[L] * 1e-3) / 1e9 / HBM_PEAK_GBS})
    else:
        out.append({"level": "coarsest solve (host LDL^T round trip, %d unknowns)" % nl, "ms": float(legs[L])})
    out.append({"level": "residual check (level 0)", "ms": float(legs[L + 1]), "algorithmic_bytes": by, "GBps": by / (legs[L + 1] * 1e-3) / 1e9,
                "frac": by / (legs[L + 1] * 1e-3) / 1e9 / HBM_PEAK_GBS})
    return {"legs": out, "sum_ms": float(legs.sum()), "note": "events at the leg boundaries cost a few us per cycle: sum_ms is above ms_per_step by that much"}


def mixed_cycle_bytes(eng, H, d):
    """Byte model of one mixed-precision step (BASELINE config 5): the inner V-cycle with s = 4 (values and vectors in fp32, int32 indices),
    the fp64 defect + norm pass over A_0 and the fp64 correction."""
    tot = 0.0
    L = eng.num_levels
    for k in range(L):
        i = eng.level_info(k); n, z = i["n"], i["nnz"]; u = H.U[k].nnz; nc = eng.level_info(k + 1)["n"]
        sweep = z * 8 + 4 * (n + 1) + 3 * n * d * 4
        tot += (eng.pre_iters + eng.post_iters + 1) * sweep + (u * 8 + 4 * (nc + 1) + n * d * 4 + nc * d * 4) + (u * 8 + 4 * (n + 1) + nc * d * 4 + 2 * n * d * 4)
    i = eng.level_info(0)
    return tot + (i["nnz"] * 12 + 4 * (i["n"] + 1) + i["n"] * d * (8 + 8 + 4) + i["n"] * 8) + i["n"] * d * (4 + 8 + 8)


def cpu_baseline(H, mass, lhs, rhs, cycles):
    """The oracle (line-by-line CPU restatement, 1 thread as the reference pins omp_set_num_threads(1),
    multigrid_solver.cpp:86-87) on the SAME workload: Galerkin setup + `cycles` V-cycles with residual check."""
    import numpy as np
    from oracle import oracle
    oracle.build()
    O = oracle.Hierarchy(H.U, mass)
    t = time.perf_counter()
    O.set_system(lhs)
    setup_s = time.perf_counter() - t
    t = time.perf_counter()
    x, it, res, conv = O.solve(rhs, tol=0.0, stop_type=2, max_iter=cycles)     # tol 0 => exactly `cycles` trips
    cyc_s = time.perf_counter() - t
    out = {
        "value": 1e3 * cyc_s / it, "unit": "ms per V-cycle (incl. residual check)", "cores": 1, "kind": "port",
        "flags": "-O3 -DNDEBUG (the reference's Release build, gravomg_bindings/setup.py:38,47: no -march)",
        "sample": f"{it} V-cycles + residual checks of the full {lhs.shape[0]}-vertex workload (x0=rhs) after the Galerkin setup",
        "setup_ms": {"reduction": O.timing["reduction"], "coarsest_solve": O.timing["coarsest_solve"], "total": 1e3 * setup_s},
        "residues": [float(r) for r in conv[:, 1]],
        "iterations_to_1e-4": int(next((i + 1 for i, r in enumerate(conv[:, 1]) if r <= 1e-4), -1)),
        "host_cpus": os.cpu_count(), "host_cpu_quota": cpu_quota(),
    }
    del O
    # the same port with -march=native, built on THIS host (a third of the cycles: it is the second baseline)
    try:
        n_cyc = max(3, cycles // 3)
        On = oracle.Hierarchy(H.U, mass, native=True)
        t = time.perf_counter(); On.set_system(lhs); nat_setup = time.perf_counter() - t
        t = time.perf_counter(); _, itn, _, convn = On.solve(rhs, tol=0.0, stop_type=2, max_iter=n_cyc); nat_s = time.perf_counter() - t
        out["march_native"] = {"value": 1e3 * nat_s / itn, "cycles": int(itn), "setup_ms_total": 1e3 * nat_setup,
                               "same_residues": bool(np.allclose(convn[:, 1], conv[:itn, 1], rtol=1e-6))}
        del On
    except Exception as e:
        out["march_native"] = {"value": None, "reason": f"{type(e).__name__}: {e}"}
    # the reference's own Eigen expressions, where this host has Eigen (oracle/eigen_baseline.cpp)
    eig, why = oracle.eigen_baseline(H.U, mass, lhs, rhs, max(3, cycles // 3))
    out["eigen"] = eig if eig is not None else {"value": None, "reason": why}
    # (oracle/eigen_baseline.cpp has never met an Eigen installation -- neither this image nor the GPU boxes ship one: unverified code, said so here)
    out["eigen"]["compiled_ever"] = eig is not None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n1", type=int, default=1732)
    ap.add_argument("--n2", type=int, default=1732)
    ap.add_argument("--order", default="natural", choices=["natural", "random", "chunks"])
    ap.add_argument("--config", default=None, choices=["1", "2", "3", "4", "4r", "4s", "5", "5b", "6"],
                    help="profiling aid: another BASELINE config (meshgen.baseline_config) as the main workload instead of the torus --n1 x --n2")
    ap.add_argument("--cpu-cycles", type=int, default=25, help="V-cycles timed on the CPU oracle (0 = skip)")
    ap.add_argument("--coarse", default="auto", choices=["auto", "host", "device"],
                    help="gmg_config::coarse_mode: auto (the library default: dense inverse applied on the device while n_L <= 8192), host (LDL^T "
                         "back-substitution, one round trip per cycle), device")
    ap.add_argument("--graph", action="store_true", help="replay the cycle legs from hipGraphs (same cycle time, ~5 ms instantiation per system)")
    ap.add_argument("--kernel-reps", type=int, default=50)
    ap.add_argument("--block-rows", type=int, default=None, help="block-hybrid GS rows per block (engine default if unset)")
    ap.add_argument("--block-from-level", type=int, default=None)
    ap.add_argument("--block-lanes", type=int, default=None)
    ap.add_argument("--no-variants", action="store_true", help="skip the informational device-coarse-apply timing")
    ap.add_argument("--force-dist", action="store_true", help="use the multi-GPU code path even with one rank")
    ap.add_argument("--exchange", default="p2p", choices=["p2p", "halo", "allgather"],
                    help="N > 1: p2p = engine-driven cycle, device-initiated stores into the peers' mailboxes (falls back to halo if it "
                         "cannot be set up); halo = RCCL all-gather of the packed halo entries per colour; allgather = whole colour segments")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong (default, the BASELINE metric) = the 3 M-vertex problem split over N ranks; weak = --n1 x --n2 vertices PER "
                         "RANK (a torus of sqrt(N) n1 x sqrt(N) n2 vertices): the weak-scaling point of SURVEY.md 8e.  Every rank still builds the whole "
                         "hierarchy and set_system (replicated set-up), so the set-up time and memory grow with N")
    ap.add_argument("--shard-levels", type=int, default=2, choices=[1, 2],
                    help="N > 1, p2p: levels partitioned over the ranks (2 = level 0 by rows per colour + level 1 by blocks; 1 = level 0 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1 or args.force_dist:
        from gravo_mg_amd import dist_bench
        return dist_bench.main(args)

    import numpy as np
    import torch
    from gravo_mg_amd import cabi

    assert torch.cuda.is_available() and cabi.device_count() > 0, "bench.py needs a HIP device (no CPU fallback)"
    workload = f"torus{args.n1}x{args.n2}-poisson-tau1e-6-d1-{args.order}"
    if args.config:
        workload, H, mass, lhs, rhs = build_config(args.config)
        args.no_variants = True
    else:
        H, mass, lhs, rhs = build_workload(args.n1, args.n2, args.order)
    n0 = lhs.shape[0]
    d0 = int(rhs.shape[1])

    kw = {}
    if args.block_rows is not None:
        kw["block_rows"] = args.block_rows
    if args.block_from_level is not None:
        kw["block_from_level"] = args.block_from_level
    if args.block_lanes is not None:
        kw["block_lanes"] = args.block_lanes
    coarse_mode = {"auto": cabi.COARSE_AUTO, "host": cabi.COARSE_HOST_LDLT, "device": cabi.COARSE_DEVICE_INVERSE}[args.coarse]
    eng = cabi.Engine(coarse_mode=coarse_mode, use_graph=args.graph, **kw)
    # A cold set-up first, on a handle that was NOT told the system's sparsity pattern (prepare_structure = 0): what a system with an
    # unannounced pattern pays (a Bilaplacian's two-ring, a caller's own prolongations) -- reported as set_system_cold_ms, never as the headline
    cold = cabi.Engine(prepare_structure=False, **kw)
    cold.use_hierarchy(H); cold.set_mass(mass)
    t = time.perf_counter(); cold.set_system(lhs); setup_cold_ms = 1e3 * (time.perf_counter() - t)
    cold.close(); del cold
    # The default: gmg_use_hierarchy hands the engine the hierarchy's point graph -- the sparsity pattern of tau M + S -- and the structure of the
    # system (orderings, colourings, layouts, symbolic Galerkin products, symbolic LDL^T) is prepared when the hierarchy is finalized
    # (structure_prepare_ms, part of the construction like the hierarchy itself); gmg_set_system then does what depends on the VALUES: upload,
    # numeric Galerkin chain, layout refill, numeric LDL^T (multigrid_solver.cpp:1387-1401)
    t = time.perf_counter()
    eng.use_hierarchy(H)
    use_hierarchy_ms = 1e3 * (time.perf_counter() - t)
    try:
        structure_ms = eng.timing("structure_prepare_ms")
    except Exception:
        structure_ms = None
    eng.set_mass(mass)
    t = time.perf_counter()
    eng.set_system(lhs)
    setup_ms = 1e3 * (time.perf_counter() - t)
    setup_prepared = bool(eng.timing("setup_structure_prepared"))
    levels = [eng.level_info(k) for k in range(eng.num_levels + 1)]
    eng_omega = eng.gs_omega
    log(f"[bench] set_system {setup_ms:.0f} ms (reduction {eng.timing('reduction'):.0f}, coarsest {eng.timing('coarsest_solve'):.0f}, "
        f"upload {eng.timing('upload'):.0f}); levels {levels}")

    # ---- solve-to-tolerance right after the setup (the reference's solve() = reduction + factorisation + loop), the other
    # half of the metric ------------------------------------------------------------------------------------
    t = time.perf_counter()
    x, iters, res, conv = eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
    solve_ms = 1e3 * (time.perf_counter() - t)
    keys = ("reduction", "coarsest_solve", "upload", "cycles", "solve_call", "solver_total", "coarse_host_ms", "solve_load", "solve_fetch")
    timing = {k: eng.timing(k) for k in keys}
    # the same solve once more on the live system (what a second right-hand side costs; also shows how much of the first call was
    # one-off: page-locking, first touch of the staging buffers, a busy host right after the set-up)
    eng.solve(rhs, tol=1e-4, stop_type=2, max_iter=100)
    timing_again = {k: eng.timing(k) for k in ("cycles", "solve_call", "coarse_host_ms", "solve_load", "solve_fetch")}

    # ---- timed region: K V-cycles (+ residual check each) on resident data ------------------------------
    eng.load_problem(rhs, rhs)                       # x0 = rhs (gravomg_bindings/src/cpp/core.cpp:69)
    eng.run_cycles(args.warmup, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    residues = eng.run_cycles(args.steps, 2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms_per_step = 1e3 * (t1 - t0) / args.steps

    # ---- roofline of the dominant kernel: fine-level Gauss-Seidel colour launches (gs_color<1,1>) --------
    sweep_ms, launches = eng.bench_kernel(0, 0, d0, args.kernel_reps)
    sweep_bytes = eng.algorithmic_bytes(0, 0, d0)
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9
    kern = {}
    for name, kind in (("residual", 1), ("restrict", 2), ("prolong_add", 3), ("norm", 4)):
        ms, _ = eng.bench_kernel(kind, 0, d0, args.kernel_reps)
        by = eng.algorithmic_bytes(kind, 0, d0)
        kern[name] = {"ms": ms, "GBps": by / (ms * 1e-3) / 1e9}
    traffic, traffic_source = load_pmc_traffic(workload)
    # level 0 stores its column indices as 16-bit codes when every slice's columns fit 8 windows: the kernel then MOVES
    # 2 bytes less per entry than the CSR figure SURVEY.md 8(d) defines as algorithmic (value + int32 index), plus 32 B of bases per
    # 64-row slice.  `achieved` / `frac` stay on the algorithmic definition; `stored_format_*` is the same launch on the bytes of the
    # stored format, i.e. how close the kernel is to the limit for what it actually has to read.
    col16 = bool(eng.timing("col16_l0"))
    info0 = eng.level_info(0)
    stored_bytes = sweep_bytes - (2.0 * (info0["nnz"] - info0["n"]) - 32.0 * info0["n_pad"] / 64 if col16 else 0.0)
    # the launch the figure is about, under the name rocprofv3 lists it by: template values <value type, right-hand sides, 1 + column-code mode (+ 2: an
    # operator that stays in the memory-side cache and is read with ordinary loads), 0>; a level 0 on the block sweep (kNN operators) is one gs_block_ep launch per sweep
    if eng.level_blocks(0) is not None:
        kernel_name = "gmgk::gs_block_ep<double,%d,...> (level 0 on the block-hybrid sweep, one launch per sweep: gmg_config::block_fine)" % d0
    else:
        fine_t = (int(eng.timing("col16_l0_mode")) + (2 if eng.timing("fine_operators_resident") else 0) + 1) if col16 else 1
        kernel_name = "gmgk::gs_color<double,%d,%d,0> (fine-level multicolour Gauss-Seidel / SOR, one launch per colour%s)" % (d0, fine_t, "; 16-bit column codes" if col16 else "")
    roofline = {
        "bound": "hbm", "kernel": kernel_name,
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_source,
        "launch_ms": sweep_ms / launches, "launches_per_sweep": launches,
        "algorithmic_bytes_per_launch": sweep_bytes / launches,
        "index_format": "16-bit window codes (2 B per entry + 32 B per slice)" if col16 else "int32",
        "stored_format_bytes_per_launch": stored_bytes / launches,
        "stored_format_frac": stored_bytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "other_fine_kernels": kern,
    }
    # ... and of the whole step (the honest companion of the dominant-kernel figure): algorithmic bytes of one V-cycle + check
    cyc_bytes = cycle_algorithmic_bytes(eng, d0)
    roofline["cycle"] = {"algorithmic_bytes": cyc_bytes, "achieved": cyc_bytes / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": cyc_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "whole V-cycle + residual check (every level, launch boundaries and the coarsest solve included) over ms_per_step"}

    if not args.graph:              # (the leg-by-leg profile needs stream launches: gmg_profile_cycle refuses a handle that replays hipGraphs)
        lv_roof = level_roofline(eng, rhs)
        roofline["levels"] = lv_roof["legs"]
        roofline["levels_sum_ms"] = lv_roof["sum_ms"]
        roofline["levels_note"] = lv_roof["note"]

    # ---- informational: the same matrix pattern with new values (the demos' new-tau-per-frame usage) only refreshes values
    lhs_b = lhs.copy()
    lhs_b.data *= 1.0 + 1e-3
    t = time.perf_counter()
    eng.set_system(lhs_b)
    repeat_ms = 1e3 * (time.perf_counter() - t)
    repeat_values_only = bool(eng.timing("setup_values_only"))

    # ---- informational variant (never `value`): the coarsest solve as the reference places it -- on the host (multigrid_solver.cpp:1075), one
    # device -> host -> device round trip per cycle -- against the default, which applies the dense inverse on the device
    variants = {}
    coarse_on_device = bool(eng.timing("coarse_on_device"))
    coarse_inverse_ms = eng.timing("coarse_inverse_ms") if coarse_on_device else None
    if args.coarse == "auto" and not args.no_variants:
        del eng
        eng2 = cabi.Engine(coarse_mode=cabi.COARSE_HOST_LDLT, use_graph=args.graph, **kw)
        eng2.use_hierarchy(H)
        eng2.set_mass(mass)
        t = time.perf_counter(); eng2.set_system(lhs); set2 = 1e3 * (time.perf_counter() - t)
        eng2.load_problem(rhs, rhs)
        eng2.run_cycles(args.warmup, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res2 = eng2.run_cycles(args.steps, 2)
        torch.cuda.synchronize()
        variants["host_coarse_solve"] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / args.steps, "set_system_ms": set2,
                                         "residues_match_default": bool(np.allclose(res2, residues, rtol=1e-6)),
                                         "what": "gmg_config::coarse_mode = GMG_COARSE_HOST_LDLT: supernodal LDL^T back-substitution on the host per cycle"}
        del eng2

    # ---- informational variants (never `value`): the same problem in random vertex order (SURVEY.md 8d asks for both orderings),
    # the reference's update without over-relaxation, and BASELINE config 3 (2 M-point kNN Laplacian)
    if not args.no_variants and args.order == "natural" and (args.n1, args.n2) == (1732, 1732):
        from gravo_mg_amd import meshgen
        variants["gs_omega_1"] = variant_run(cabi, torch, "level-0 omega = 1 (the reference's update in colour order)", H, mass, lhs, rhs, args.steps, args.warmup, gs_omega=1.0)
        Hr, mass_r, lhs_r, rhs_r = build_workload(args.n1, args.n2, "random")
        variants["random_vertex_order"] = variant_run(cabi, torch, "random vertex order", Hr, mass_r, lhs_r, rhs_r, args.steps, args.warmup)
        del Hr, mass_r, lhs_r, rhs_r
        # an ordering in between: 65 536-vertex runs of the natural order, shuffled -- local inside a run, below the reordering trigger
        Hc, mass_c, lhs_c, rhs_c = build_workload(args.n1, args.n2, "chunks")
        variants["chunk_shuffled_order"] = variant_run(cabi, torch, "natural order in shuffled runs of 65 536 vertices (not renumbered)", Hc, mass_c, lhs_c, rhs_c, args.steps, args.warmup)
        del Hc, mass_c, lhs_c, rhs_c
        # the reference's real call pattern (demos/smoothing.py:47-50, conformal_flow.py:58): n x 3 right-hand side, lhs = M + 1e-3 S,
        # on the same 3 M-vertex mesh and hierarchy; byte model 160 B per fine row and sweep instead of 112
        Vs, Fs = meshgen.torus_mesh(args.n1, args.n2)
        Ss, mass_s = meshgen.cotan_laplacian(Vs, Fs)
        lhs_s, rhs_s = meshgen.smoothing_system(Ss, mass_s, Vs)
        variants["smoothing_d3_3M"] = variant_run(cabi, torch, "smoothing M + 1e-3 S, d = 3", H, mass, lhs_s, rhs_s, args.steps, args.warmup, kernels=True, levels=True)
        variants["smoothing_d3_3M"]["ratio_to_d1_cycle"] = variants["smoothing_d3_3M"]["ms_per_step"] / ms_per_step
        # BASELINE config 5: Bilaplacian data smoothing M + tau S M^-1 S on the same mesh, fp64 and with the fp32 inner V-cycle.  tau = 1e-9: with the
        # reference's 1e-3 the reference iteration itself does not contract at this size; neither precision reaches 1e-4 within
        # max_iter = 100, so the count is to 3e-2
        lhs_b5, rhs_b5 = meshgen.smoothing_system(meshgen.bilaplacian(Ss, mass_s), mass_s, Vs[:, :1], tau=1e-9)
        del Vs, Fs, Ss, lhs_s, rhs_s
        variants["bilaplacian_3M"] = {
            "system": "M + 1e-9 S M^-1 S, d = 1, 19 entries per row",
            "fp64": variant_run(cabi, torch, "Bilaplacian fp64", H, mass, lhs_b5, rhs_b5, args.steps, args.warmup, reach=3e-2, levels=True),
            "fp32_inner": variant_run(cabi, torch, "Bilaplacian fp32 inner cycle", H, mass, lhs_b5, rhs_b5, args.steps, args.warmup, reach=3e-2, inner_precision=1)}
        del lhs_b5, rhs_b5, mass_s
        # BASELINE config 2: ~720 k-vertex cotangent Poisson
        name2, pos2, S2, mass2, lhs2, rhs2 = meshgen.baseline_config("2")
        H2 = cabi.Hierarchy(pos2, meshgen.neighbors_from_stiffness(S2), ratio=8.0, lower_bound=1000)
        variants["poisson_722k"] = variant_run(cabi, torch, name2, H2, mass2, lhs2, rhs2, args.steps, args.warmup, levels=True)
        del H2, pos2, S2, mass2, lhs2, rhs2
        # the demos' own size (demos/smoothing.py on a ~36 k-vertex mesh): M + 1e-3 S, n x 3 -- a cycle here is launch latency and the host's
        # coarsest solve (6 k unknowns), not bytes; the dense coarsest inverse on the device beside it (what it costs to MAKE is in its set_system_ms)
        Vm, Fm = meshgen.torus_mesh(190, 190)
        Sm, mass_m = meshgen.cotan_laplacian(Vm, Fm)
        lhs_m, rhs_m = meshgen.smoothing_system(Sm, mass_m, Vm)
        Hm = cabi.Hierarchy(Vm, meshgen.neighbors_from_stiffness(Sm), ratio=8.0, lower_bound=1000)
        variants["smoothing_36k_d3"] = variant_run(cabi, torch, "smoothing M + 1e-3 S, d = 3, 36 100 vertices", Hm, mass_m, lhs_m, rhs_m, args.steps, args.warmup)
        dc = variant_run(cabi, torch, "smoothing 36 k, d = 3, coarsest solve on the host", Hm, mass_m, lhs_m, rhs_m, args.steps, args.warmup, coarse_mode=cabi.COARSE_HOST_LDLT)
        variants["smoothing_36k_d3"]["host_coarse_solve"] = {k: dc[k] for k in ("ms_per_step", "iterations_to_1e-4", "solve_ms", "second_solve_ms", "set_system_ms")}
        del Vm, Fm, Sm, mass_m, lhs_m, rhs_m, Hm
        name, pos, S3, mass3, lhs3, rhs3 = meshgen.baseline_config("3")
        H3 = cabi.Hierarchy(pos, meshgen.neighbors_from_stiffness(S3), ratio=8.0, lower_bound=1000)
        variants["pointcloud_2M_knn8"] = variant_run(cabi, torch, name, H3, mass3, lhs3, rhs3, args.steps, args.warmup, levels=True)
        # the same system with level 0 kept colour-major (what rounds 1-3 ran, and what the partitioned multi-GPU cycle runs): 11 launches per sweep
        cm = variant_run(cabi, torch, name + " (block_fine = 0)", H3, mass3, lhs3, rhs3, args.steps, args.warmup, block_fine=0)
        variants["pointcloud_2M_knn8"]["level0_colour_major"] = {k: cm[k] for k in ("ms_per_step", "iterations_to_1e-4", "solve_ms", "second_solve_ms", "set_system_ms", "colors", "cycle_frac_of_peak")}
        del H3, pos, S3, mass3, lhs3, rhs3

    cpu = cpu_baseline(H, mass, lhs, rhs, args.cpu_cycles) if args.cpu_cycles > 0 else None

    out = {
        "metric": "V-cycle wall time (ms per V-cycle incl. residual check) + solve-to-1e-4 iterations, 3M-vertex Poisson",
        "value": ms_per_step, "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "n_vertices": n0, "levels": [l["n"] for l in levels], "colors": [l["n_colors"] for l in levels],
                   "smoother": f"2+2 sweeps; level 0: multicolour Gauss-Seidel over-relaxed by {eng_omega:g} (SOR, one launch per colour); levels >= 1: "
                               "block-hybrid Gauss-Seidel (64-row blocks: Gauss-Seidel inside a block, Jacobi between blocks, one launch per sweep)",
                   "coarse_solve": args.coarse + (" (dense inverse applied on the device, built on the device at set_system)" if coarse_on_device else " (host LDL^T per cycle)"),
                   "hipgraph": args.graph,
                   "tolerance": 1e-4, "stopping_criteria": 2},
        "iterations_to_1e-4": iters, "residue": res, "residues_to_1e-4": [float(v) for v in conv[:, 1]],
        "iterations_reference_algorithm": cpu["iterations_to_1e-4"] if cpu else None,
        "solve_ms": solve_ms, "solver_timing_ms": timing, "second_solve_timing_ms": timing_again,
        "set_system_ms": setup_ms, "set_system_structure_prepared": setup_prepared, "structure_prepare_ms": structure_ms, "use_hierarchy_ms": use_hierarchy_ms,
        "set_system_cold_ms": setup_cold_ms, "coarse_inverse_ms": coarse_inverse_ms,
        "construct_plus_first_solve_ms": use_hierarchy_ms + setup_ms + solve_ms,
        "set_system_note": "set_system_ms: the first gmg_set_system on a handle whose hierarchy announced the system's sparsity pattern (its point graph): "
                           "values up, numeric Galerkin chain, layout refill, numeric LDL^T; the structural half was done once in gmg_use_hierarchy "
                           "(structure_prepare_ms, inside use_hierarchy_ms).  set_system_cold_ms: the same call on a handle that was told nothing",
        "set_system_same_pattern_ms": repeat_ms, "set_system_same_pattern_values_only": repeat_values_only,
        "mvertex_cycles_per_s": n0 / ms_per_step / 1e3,
        "timed_residues_tail": [float(r) for r in residues[-3:]],
        "variants": variants,
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if cpu:
        out["speedup_vs_cpu_per_cycle"] = cpu["value"] / ms_per_step
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
