"""`gravomg.MultigridSolver` -- the reference's Python API (gravomg_bindings/src/gravomg/core.py:7-147) over the
MI355X-native V-cycle engine.  Same constructor keywords and defaults, same methods and properties; the numerics
run in libgravomg_hip.so (HIP kernels) through the pybind11 module `gravomg_bindings` next to this package.

Differences a user can observe, all deliberate:
  * only the default hierarchy (Sampling.FASTDISK) and the V-cycle (cycle_type=0) exist; asking for SIG06 / SIG21 /
    ablation hierarchies, other samplers or F-/W-cycles raises instead of printing a message;
  * problems (no GPU, unsupported option, singular coarse operator) raise RuntimeError instead of being printed;
  * the smoother is multicolour Gauss-Seidel (the reference's sweep in a colour-permuted order): iteration counts
    match the reference's on the tested problems, per-cycle iterates are not bitwise those of lexicographic GS.
"""
import numpy as np
from scipy.sparse import csr_matrix

def _preload_torch_hip_runtime():
    # torch-ROCm wheels bundle libamdhip64 / libhsa-runtime64 with the system SONAMEs; whichever copy loads first serves
    # the whole process and torch fails if it is not its own.  Make torch's copy (if torch is installed) the one in use.
    import ctypes, importlib.util, os
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


_preload_torch_hip_runtime()

import gravomg_bindings  # noqa: E402
from gravomg_bindings import Hierarchy, Sampling, Weighting  # noqa: E402


class MultigridSolver(object):
    def __init__(
        self, pos, neigh, mass,
        ratio=8.0, lower_bound=1000, cycle_type=0, tolerance=1e-4, stopping_criteria=2, pre_iters=2, post_iters=2, max_iter=100,
        check_voronoi=True, nested=False, sampling_strategy=Sampling.FASTDISK, weighting=Weighting.BARYCENTRIC,
        sig06=False, normals=None, verbose=False, debug=False, ablation=False, ablation_num_points=3, ablation_random=False,
    ):
        """Builds the Gravo MG hierarchy for a mesh or point cloud and prepares the solver.

        pos: (n, 3) positions.  neigh: (n, K) int neighbour table padded with -1 (see gravomg.util).
        mass: lumped (diagonal) scipy sparse mass matrix.  The remaining keywords are the reference's
        (ratio, lower_bound: hierarchy; cycle_type, tolerance, stopping_criteria -- 2 is the M-weighted
        residual norm --, pre_iters, post_iters, max_iter: solver)."""
        super().__init__()
        if not mass.getformat() == 'csr':
            mass = mass.tocsr()
        pos = np.asarray(pos, dtype=np.float64)
        normals = pos if normals is None else normals
        self.solver = gravomg_bindings.MultigridSolver(
            pos, np.asarray(neigh, dtype=np.int32), mass,
            ratio, lower_bound, cycle_type, tolerance, stopping_criteria, pre_iters, post_iters, max_iter,
            check_voronoi, nested, sampling_strategy, weighting,
            sig06, normals, verbose, debug, ablation, ablation_num_points, ablation_random,
        )
        self.sig21_computed = False
        self.sig21bary_computed = False

    def construct_sig21_hierarchy(self, faces):
        """Liu et al. [2021] comparison hierarchy: out of scope here, raises."""
        self.solver.construct_sig21_hierarchy(faces)

    def toggle_hierarchy(self, hierarchy_type):
        assert hierarchy_type == Hierarchy.OURS or (hierarchy_type == Hierarchy.SIG21 and self.sig21_computed)
        self.solver.toggle_hierarchy(hierarchy_type)

    def solve(self, lhs, rhs):
        """Solves lhs @ x = rhs with V-cycles from the initial guess x0 = rhs; returns x as an (n, d) array."""
        if not lhs.getformat() == 'csr':
            print('LHS is not in CSR format, converting to CSR')
            lhs = lhs.tocsr()
        return self.solver.solve(lhs, rhs)

    def direct_solve(self, lhs, rhs, pardiso=False):
        """Direct sparse LDL^T on the host (comparison helper; Pardiso is not available)."""
        return self.solver.direct_solve(lhs, rhs, pardiso)

    # Getters and setters

    @property
    def prolongation_matrices(self):
        return self.solver.prolongation_matrices()

    def set_prolongation_matrices(self, U):
        self.solver.set_prolongation_matrices(list(U))

    @property
    def sampling_indices(self):
        return self.solver.sampling_indices()

    @property
    def level_points(self):
        return self.solver.level_points()

    @property
    def level_edges(self):
        return self.solver.level_edges()

    @property
    def notrimap(self):
        return self.solver.notrimap()

    @property
    def all_triangles(self):
        return self.solver.all_triangles()

    @property
    def coarse_normals(self):
        return self.solver.coarse_normals()

    @property
    def nearest_source(self):
        return self.solver.nearest_source()

    # Timing logs (same CSV layout as the reference's writers)

    def write_hierarchy_timing(self, experiment, file, write_headers=False):
        return self.solver.write_hierarchy_timing(experiment, file, write_headers)

    def write_solver_timing(self, experiment, file, write_headers=False):
        return self.solver.write_solver_timing(experiment, file, write_headers)

    def write_convergence(self, file):
        return self.solver.write_convergence(file)

    def residual(self, lhs, rhs, solution, type=2):
        return self.solver.residual(lhs, rhs, solution, type)

    # Extras of this build (not upstream)

    @property
    def solver_timing(self):
        """dict with the reference's solverTiming keys: reduction, coarsest_solve, cycles, solver_total, iterations, residue."""
        return self.solver.solver_timing()

    @property
    def hierarchy_timing(self):
        return self.solver.hierarchy_timing()

    @property
    def convergence(self):
        """[(elapsed_ms, residue), ...] per V-cycle (appended across solves, like upstream)."""
        return self.solver.convergence()

    def set_engine_option(self, key, value):
        """MI355X engine knobs: smoother (0 multicolour GS, 1 Jacobi), jacobi_omega, coarse_mode (0 host LDL^T, 1 device),
        use_graph, block_rows, block_from_level, device."""
        self.solver.set_engine_option(key, float(value))
