"""`gravomg.MultigridSolver`: the Python class name, constructor keywords, defaults, methods and properties of the
reference package (the API contract: gravomg_bindings/src/gravomg/core.py:7-147 and the pybind class it wraps,
gravomg_bindings/src/cpp/core.cpp:142-163) on top of the MI355X-native V-cycle engine.  The numerics run in
libgravomg_hip.so (HIP kernels, C-ABI in include/gravomg_hip.h) through the pybind11 module `gravomg_bindings` that sits
next to this package; this file only normalises arguments and forwards.

What a user of the reference will notice, all of it deliberate (DESIGN.md sections 2 and 7):
  * only the default hierarchy (Sampling.FASTDISK) and the V-cycle (cycle_type=0) exist.  SIG06 / SIG21 / ablation
    hierarchies, the other samplers, F-/W-cycles and Pardiso raise instead of printing a message;
  * problems (no GPU, unsupported option, singular coarsest operator) raise RuntimeError instead of being printed;
  * the smoother is a parallel ordering of the reference's Gauss-Seidel: multicolour sweeps on the finest level, over-relaxed
    by 1.35 (`set_engine_option("gs_omega", 1.0)` gives the reference's update in colour order), block sweeps below.  The
    iterates are therefore not those of lexicographic Gauss-Seidel cycle by cycle; the stopping test and the solution are the
    same.  V-cycles to 1e-4 on the 3 M-vertex Poisson problem: 4 (reference algorithm: 6; omega = 1: 7).  Should these smoothers
    diverge on some matrix, solve() repeats the solve with Gauss-Seidel (colour order) on every level and says so in
    solver_timing["fallback_exact_gs"];
  * `lhs` may be any scipy sparse format.  CSR and symmetric CSC storage are used in place (no conversion, no copy).
"""
import numpy as np
import scipy.sparse as sp


def _use_torch_hip_runtime():
    """torch-ROCm wheels ship their own libamdhip64 / libhsa-runtime64 under the system SONAMEs; the first copy loaded serves
    the process and torch refuses to see a GPU if it is not its own.  If torch is installed, load its copy first."""
    import ctypes
    import importlib.util
    import os
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    roots = list(spec.submodule_search_locations or []) if spec is not None else []
    for root in roots[:1]:
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(root, "lib", name)
            if os.path.exists(path):
                try:
                    ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
                except OSError:
                    pass


_use_torch_hip_runtime()

import gravomg_bindings as _native  # noqa: E402
from gravomg_bindings import Hierarchy, Sampling, Weighting  # noqa: E402,F401


def _sparse(m, what):
    if not sp.issparse(m):
        raise TypeError(f"{what} must be a scipy sparse matrix")
    return m


def _points(a, what, cols=None):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    if a.ndim != 2 or (cols is not None and a.shape[1] != cols):
        raise ValueError(f"{what} must be an (n, {cols or 'd'}) array")
    return a


# name -> (kind, docstring).  "call": method forwarding its arguments; "get": read-only property over a native getter.
_FORWARDED = {
    "construct_sig21_hierarchy": ("call", "Comparison hierarchy of Liu et al. 2021: out of scope of this build, raises."),
    "direct_solve": ("call", "direct_solve(lhs, rhs, pardiso=False): sparse LDL^T on the host (comparison helper); pardiso=True raises."),
    "write_hierarchy_timing": ("call", "write_hierarchy_timing(experiment, file, write_headers=False): the reference's CSV layout."),
    "write_solver_timing": ("call", "write_solver_timing(experiment, file, write_headers=False): the reference's CSV layout."),
    "write_convergence": ("call", "write_convergence(file): one 'time,residue' line per V-cycle."),
    "prolongation_matrices": ("get", "List of the prolongation operators U_k (scipy CSC, n_k x n_{k+1})."),
    "sampling_indices": ("get", "sampling_indices[k][c]: index on level k of point c of level k + 1."),
    "nearest_source": ("get", "nearest_source[k][i]: the level-(k+1) point whose cluster contains point i of level k."),
    "level_points": ("get", "Positions of the points of every coarse level ((n_k, 3) arrays); filled when debug=True, like upstream."),
    "level_edges": ("get", "Edges of the SIG06 hierarchy's levels: empty for the default hierarchy, as upstream."),
    "notrimap": ("get", "With debug=True one zero vector per level (upstream never writes anything else into it); empty otherwise."),
    "all_triangles": ("get", "With debug=True the candidate triangles of every level's coarse points; empty otherwise."),
    "coarse_normals": ("get", "Never filled upstream: always empty."),
    "solver_timing": ("get", "dict with the reference's solverTiming keys: reduction, coarsest_solve, cycles, solver_total, iterations, residue."),
    "hierarchy_timing": ("get", "dict with the reference's hierarchyTiming keys."),
    "convergence": ("get", "[(elapsed_ms, residue), ...] per V-cycle, appended across solves like upstream."),
}
# parameter names of the forwarded methods, and their defaults where the reference has some
_PARAMS = {
    "construct_sig21_hierarchy": (("faces",), {}),
    "direct_solve": (("lhs", "rhs", "pardiso"), {"pardiso": False}),
    "write_hierarchy_timing": (("experiment", "file", "write_headers"), {"write_headers": False}),
    "write_solver_timing": (("experiment", "file", "write_headers"), {"write_headers": False}),
    "write_convergence": (("file",), {}),
}


class MultigridSolver(object):
    """Gravo MG solver for a mesh or point cloud: builds the Graph-Voronoi hierarchy at construction, solves with V-cycles.

    pos (n, 3) positions; neigh (n, K) integer neighbour table padded with -1 (gravomg.util); mass: lumped (diagonal) scipy
    sparse mass matrix.  ratio / lower_bound steer the hierarchy; cycle_type, tolerance, stopping_criteria (2 = M-weighted
    residual norm), pre_iters, post_iters, max_iter the solver.  Keywords and defaults are the reference's."""

    def __init__(self, pos, neigh, mass,
                 ratio=8.0, lower_bound=1000, cycle_type=0, tolerance=1e-4, stopping_criteria=2, pre_iters=2, post_iters=2, max_iter=100,
                 check_voronoi=True, nested=False, sampling_strategy=Sampling.FASTDISK, weighting=Weighting.BARYCENTRIC,
                 sig06=False, normals=None, verbose=False, debug=False, ablation=False, ablation_num_points=3, ablation_random=False):
        pos = _points(pos, "pos", 3)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        if neigh.ndim != 2 or neigh.shape[0] != pos.shape[0]:
            raise ValueError("neigh must be an (n, K) integer table with one row per point")
        self.solver = _native.MultigridSolver(
            pos, neigh, _sparse(mass, "mass"), float(ratio), int(lower_bound), int(cycle_type), float(tolerance), int(stopping_criteria),
            int(pre_iters), int(post_iters), int(max_iter), bool(check_voronoi), bool(nested), sampling_strategy, weighting, bool(sig06),
            pos if normals is None else _points(normals, "normals", 3), bool(verbose), bool(debug), bool(ablation), int(ablation_num_points),
            bool(ablation_random))

        self._solve_args = (float(tolerance), int(stopping_criteria), int(max_iter))
        self._dist = None

    def solve(self, lhs, rhs):
        """x with lhs @ x = rhs to the tolerance given at construction, V-cycles from the initial guess x0 = rhs
        (gravomg_bindings/src/cpp/core.cpp:68-72).  rhs (n,) or (n, d); returns an (n, d) array.  After enable_distributed() the
        V-cycles run row-partitioned over the ranks (same iterates, same result on every rank)."""
        if self._dist is not None:
            return self._solve_distributed(_sparse(lhs, "lhs"), _points(rhs, "rhs"))
        return self.solver.solve(_sparse(lhs, "lhs"), _points(rhs, "rhs"))

    # ---- one process per GPU (not upstream) -------------------------------------------------------------------------------------------
    def enable_distributed(self, rank, world, all_gather, device=None, shard_levels=2, partition_setup=True, exchange="mailbox"):
        """Make solve() a COLLECTIVE over `world` processes (one per GPU), each holding a MultigridSolver built from the same inputs:
        level 0 is partitioned by rows per colour (levels >= 1 by blocks / replicated, `shard_levels`), exchanges are device-initiated
        stores into the peers' mailboxes (include/gravomg_hip.h, "multi-GPU, engine-driven").  The colours are
        global, so the iterates -- and the returned x, on every rank -- are those of the single-GPU solve.

        all_gather(obj) -> list with every rank's obj in rank order, e.g.
            def all_gather(o): out = [None] * world; torch.distributed.all_gather_object(out, o); return out
        (the only thing the ranks exchange through the caller: 1 KB connection records, once per system layout).
        device: HIP device of this rank (default: rank).  Call before the first solve().
        partition_setup (default): the engine lays out and keeps only this rank's rows of levels 0-1 (gmg_dist_partition: a rank's device
        memory is its share of the operator plus the replicated small levels; not for systems whose level 0 runs the block sweep -- kNN graph
        Laplacians --, which keep the single-GPU smoother and take whole operators on every rank); such an object runs the collective solve() only --
        residual() and the single-process entry points need partition_setup=False (every rank then holds the whole operator).
        exchange: "mailbox" (default: device-initiated stores through hipIpc mappings, one launch per exchange) or "rccl" (every exchange as
        pack -> ncclAllGather -> unpack on the engine's stream, gmg_config::dist_exchange = 1: for boxes where processes cannot map each
        other's device memory); the same partition, the same iterates.  (The coarsest solve stays where the single-process object has it -- on the host by
        default -- so that the ranks reproduce its iterates bit for bit; set_engine_option("coarse_mode", 2) before the first solve moves it onto
        the devices: no host work per rank inside a cycle.)"""
        if exchange not in ("mailbox", "rccl"):
            raise ValueError("exchange must be 'mailbox' or 'rccl'")
        rank, world = int(rank), int(world)
        if not (0 <= rank < world):
            raise ValueError("rank must be in [0, world)")
        if world > 1:
            self.solver.set_engine_option("row_align", 64 * world)
            self.solver.set_engine_option("dist_shard_levels", int(shard_levels))
            self.solver.set_engine_option("device", rank if device is None else int(device))
            self.solver.set_engine_option("dist_rank", rank if partition_setup else 0)
            self.solver.set_engine_option("dist_world", world if partition_setup else 1)
            self.solver.set_engine_option("dist_exchange", 1 if exchange == "rccl" else 0)
            self._dist = {"rank": rank, "world": world, "all_gather": all_gather, "cycle": None, "key": None, "exchange": exchange, "partition_setup": bool(partition_setup), "rule_checked": False}
        else:
            self._dist = None

    def _solve_distributed(self, lhs, rhs):
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        if root not in sys.path:
            sys.path.insert(0, root)
        from gravo_mg_amd import cabi                     # ctypes view of the same libgravomg_hip.so
        D = self._dist
        tol, stop_type, max_iter = self._solve_args
        if D["partition_setup"] and not D["rule_checked"]:
            # One smoother per system at every N: an operator whose level 0 runs the block sweep on one GPU (long rows, Stieltjes signs: kNN graph
            # Laplacians, gmg_config::block_fine) runs it on N too, partitioned by runs of whole blocks -- but a partitioned SET-UP keeps level 0
            # colour-major, so such a system takes the whole-operator set-up on every rank.  The rule looks at the matrix only: every rank decides alike.
            D["rule_checked"] = True
            if cabi.host_fine_block_rule(lhs)[0]:
                D["partition_setup"] = False
                self.solver.set_engine_option("dist_rank", 0)
                self.solver.set_engine_option("dist_world", 1)
        handle, generation = self.solver.prepare_system(lhs)
        key = (handle, generation, rhs.shape[1])
        if D["key"] != key:                               # new layout (or another d): partition, export, connect
            eng = cabi.Engine.borrow(handle)
            cyc = cabi.P2PCycle(eng, D["rank"], D["world"], rhs.shape[1])
            if D["exchange"] == "rccl":
                ids = D["all_gather"](cabi.rccl_unique_id() if D["rank"] == 0 else None)      # rank 0 makes the communicator id
                cyc.connect_rccl(ids[0])
            else:
                cyc.connect(D["all_gather"](cyc.export()))
            D["cycle"], D["key"] = cyc, key
        # x0 = rhs, as the binding does; then gmg_p2p_solve: do { V-cycle; residualCheck } while (residue > tol && it < maxIter)
        x, it, residue = D["cycle"].solve(rhs, rhs, tol=tol, stop_type=stop_type, max_iter=max_iter)
        diverged = bool(D["cycle"].diverged)
        self.distributed_info = {"iterations": it, "residue": residue, "world": D["world"], "diverged": diverged}
        if diverged:
            # every rank sees the same residues, so every rank lands here together.  The single-GPU path would repeat the solve with Gauss-Seidel
            # on every level; here the caller is told (as the reference would not be) and a blown-up iteration is an error, never a result
            print(f"gravomg: the V-cycle iteration did not contract over {D['world']} ranks (residue {residue} after {it} cycles); x holds the last iterate; "
                  "set_engine_option('block_rows', 0) and ('gs_omega', 1.0) select Gauss-Seidel in colour order on every level", flush=True)
            if not np.isfinite(residue) or D["cycle"].eng.timing("blown_up"):
                raise RuntimeError(f"the V-cycle iteration diverged (residue {residue})")
        return np.ascontiguousarray(x)

    def residual(self, lhs, rhs, solution, type=2):
        """The reference's residualCheck of `solution`: type 0 ||r||/||b||, 1 M^-1-weighted, 2 M-weighted, 3 ||A X - B||_F."""
        return self.solver.residual(_sparse(lhs, "lhs"), _points(rhs, "rhs"), _points(solution, "solution"), int(type))

    def set_prolongation_matrices(self, U):
        """Replace the hierarchy by the given prolongation operators (scipy sparse, n_k x n_{k+1})."""
        self.solver.set_prolongation_matrices([_sparse(u, "U[k]") for u in U])

    def toggle_hierarchy(self, hierarchy_type):
        """Only Hierarchy.OURS exists in this build; anything else raises."""
        self.solver.toggle_hierarchy(hierarchy_type)

    def set_engine_option(self, key, value):
        """MI355X engine knobs (not upstream): smoother (0 multicolour Gauss-Seidel, 1 weighted Jacobi), gs_omega, jacobi_omega,
        coarse_mode (0 host LDL^T back-substitution per cycle, 1 dense inverse built and applied on the device, 2 = default: 1 while the coarsest level has <= 8 192 unknowns -- a system solved ONCE is faster with 0), use_graph, block_rows, block_from_level, device."""
        self.solver.set_engine_option(str(key), float(value))


def _bind(name, args, kwargs):
    """Positional argument tuple for the native method `name` from the caller's args / kwargs (reference names and defaults)."""
    names, defaults = _PARAMS[name]
    if len(args) > len(names):
        raise TypeError(f"{name}() takes {len(names)} arguments ({len(args)} given)")
    bound = dict(defaults)
    bound.update(zip(names, args))
    for k, v in kwargs.items():
        if k not in names or k in names[:len(args)]:
            raise TypeError(f"{name}() got an unexpected or repeated argument '{k}'")
        bound[k] = v
    missing = [n for n in names if n not in bound]
    if missing:
        raise TypeError(f"{name}() missing argument(s): {', '.join(missing)}")
    return tuple(bound[n] for n in names)


def _install_forwarders():
    for name, (kind, doc) in _FORWARDED.items():
        if kind == "get":
            setattr(MultigridSolver, name, property(lambda self, _n=name: getattr(self.solver, _n)(), doc=doc))
            continue

        def method(self, *args, _n=name, **kwargs):
            return getattr(self.solver, _n)(*_bind(_n, args, kwargs))
        method.__name__ = method.__qualname__ = name
        method.__doc__ = doc
        setattr(MultigridSolver, name, method)


_install_forwarders()
